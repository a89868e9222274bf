set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export AGPT_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --spp 8 > gpurun_out/rehearse2.log 2> gpurun_out/rehearse2.err || { tail -30 gpurun_out/rehearse2.err; exit 1; }
cat gpurun_out/rehearse2.log | cut -c1-900
unset AGPT_BENCH_BACKEND
python bench.py --steps 1 --warmup 0 --spp 16 --no-cpu-baseline | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1 16spp mean', d['image_mean_rgb'], d['value'])"
