set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/final/tests.log 2>&1 || { tail -40 gpurun_out/final/tests.log; exit 1; }
tail -2 gpurun_out/final/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1 || { cat gpurun_out/final/smoke.log; exit 1; }
cat gpurun_out/final/smoke.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { tail -30 gpurun_out/final/bench.err; exit 1; }
cat gpurun_out/final/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final -o r01 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/final/prof_bench.log 2> gpurun_out/final/prof_bench.err
head -5 gpurun_out/final/r01_kernel_stats.csv | cut -c1-80
bash tools/gpu_traffic.sh > gpurun_out/final/traffic.log 2>&1 || { tail gpurun_out/final/traffic.log; exit 1; }
tail -3 gpurun_out/final/traffic.log | cut -c1-200
