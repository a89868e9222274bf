set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { cat gpurun_out/smoke.log; exit 1; }
cat gpurun_out/smoke.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.log 2> gpurun_out/bench1.err || { tail -30 gpurun_out/bench1.err; exit 1; }
cat gpurun_out/bench1.log
