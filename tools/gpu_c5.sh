set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/render.py --scene c5 --width 960 --height 540 --spp 32 --out gpurun_out/c5.png
python bench.py --workload c5 --steps 1 --warmup 1 --spp 16 --no-cpu-baseline > gpurun_out/bench_c5.log 2> gpurun_out/bench_c5.err || { tail -30 gpurun_out/bench_c5.err; exit 1; }
cat gpurun_out/bench_c5.log
