set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/traffic
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/traffic -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/traffic/$name.log 2> gpurun_out/traffic/$name.err || { tail -20 gpurun_out/traffic/$name.err; exit 1; }; }
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/traffic_json.py gpurun_out/traffic gpurun_out/traffic/traffic.json
# full-size kernel trace + stats for profiles/
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r01 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_bench.log 2> gpurun_out/prof_bench.err
head -4 gpurun_out/prof/r01_kernel_stats.csv | cut -c1-200
