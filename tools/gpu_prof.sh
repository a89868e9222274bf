set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.log 2> gpurun_out/prof_bench.err || { tail -30 gpurun_out/prof_bench.err; exit 1; }
cat gpurun_out/prof_bench.log | cut -c1-400
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
cat $f | head -20
