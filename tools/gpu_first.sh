set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s > gpurun_out/first_tests.log 2>&1 || { tail -60 gpurun_out/first_tests.log; exit 1; }
tail -30 gpurun_out/first_tests.log
