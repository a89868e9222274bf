# TCP / TA pressure probe of the trace kernels (developer tool): small workload, one counter group per pass, progress lines
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/tcp
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; echo "pass $name ..."; timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $name -- python3 bench.py --steps 1 --warmup 0 --spp 8 --no-cpu-baseline --no-secondary > $O/$name.log 2> $O/$name.err || { echo "pass $name failed"; tail -5 $O/$name.err; return 0; }; echo "pass $name done"; }
run tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp3 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run ta TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
python3 tools/pmc_agg.py $O/*_counter_collection.csv > $O/agg.txt 2>&1 || true
grep -A12 "k_trace_fast<0" $O/agg.txt | head -80
