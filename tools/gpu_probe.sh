# round-2 probe: phase clocks of k_trace_fast (clock variant), TA/TCP counters, baseline bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
export TMPDIR=/tmp
O=gpurun_out/probe
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/base.log 2> $O/base.err || { tail -20 $O/base.err; exit 1; }
cut -c1-250 $O/base.log
AGPT_LIB_VARIANT=clock python bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/clock.log 2> $O/clock.err || { tail -20 $O/clock.err; exit 1; }
grep "trace " $O/clock.err | tail -6
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $name -- python3 bench.py --steps 1 --warmup 0 --spp 8 --no-cpu-baseline > $O/$name.log 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }; echo "pmc $name done"; }
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum
run tcp2 TCP_TA_TCP_STATE_READ_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum
run tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
python3 tools/pmc_agg.py $O/*_counter_collection.csv > $O/agg.txt 2>&1 || true
tail -60 $O/agg.txt
