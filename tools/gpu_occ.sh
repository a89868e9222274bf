set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AGPT_LIB_VARIANT=s23 AGPT_BLOCKS_PER_CU=5 python -m pytest tests/test_gpu_render.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
bash tools/gpu_ab.sh "AGPT_LIB_VARIANT=s23 AGPT_BLOCKS_PER_CU=4" "AGPT_LIB_VARIANT=s23 AGPT_BLOCKS_PER_CU=5" "AGPT_LIB_VARIANT=s23 AGPT_BLOCKS_PER_CU=6"
