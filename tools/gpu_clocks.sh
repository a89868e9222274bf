# Samples power and shader clock while the C3 bench runs: bash tools/gpu_clocks.sh -> gpurun_out/clocks.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/clk_bench.log 2> gpurun_out/clk_bench.err &
BP=$!

for i in $(seq 1 150); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|power\|junction\|mclk" | tr '\n' ' ' ; echo
  sleep 0.15
  kill -0 $BP 2>/dev/null || break
done > gpurun_out/clocks.log
wait $BP
tail -c 600 gpurun_out/clk_bench.log | head -c 300; echo
awk "{print NR\": \"\$0}" gpurun_out/clocks.log | sed -e "s/GPU\[0\]\t\t: //g" -e "s/=*//g" | cut -c1-200 | awk "NR%4==0" | head -45
