# Vector-L1 side counters of one C3 bench step: bash tools/gpu_pmc_l1.sh  ->  gpurun_out/pmc_l1/ (one small --pmc group per pass,
# --kernel-trace only; every pass under its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and linger)
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_l1
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; echo "pass $name: $@"; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/$name.log 2> $O/$name.err || { grep -m2 "error code\|Could not" $O/$name.err; echo "pass $name failed"; }; }
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
run ta3 TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
run grbm GRBM_GUI_ACTIVE
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
run tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run tcp5 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run tcp6 TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
run tcp7 TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum
python3 tools/pmc_l1_json.py $O "c3:1920x1080:64spp:world1:scale1" gpurun_out/pmc_r03_c3_l1.json
