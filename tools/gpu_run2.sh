set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/tests2.log 2>&1 || { tail -40 gpurun_out/tests2.log; exit 1; }
tail -3 gpurun_out/tests2.log
python bench.py --steps 2 --warmup 1 > gpurun_out/bench2.log 2> gpurun_out/bench2.err || { tail -30 gpurun_out/bench2.err; exit 1; }
cat gpurun_out/bench2.log
