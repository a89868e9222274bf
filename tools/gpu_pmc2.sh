set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc2 -o $name -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline > gpurun_out/pmc2/$name.log 2> gpurun_out/pmc2/$name.err || { tail -20 gpurun_out/pmc2/$name.err; exit 1; }; echo "== $name"; python3 tools/pmc_agg.py gpurun_out/pmc2/${name}_counter_collection.csv; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
