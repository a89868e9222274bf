set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc3 -o $name -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline > gpurun_out/pmc3/$name.log 2> gpurun_out/pmc3/$name.err || { tail -20 gpurun_out/pmc3/$name.err; exit 1; }; echo "== $name"; python3 tools/pmc_agg.py gpurun_out/pmc3/${name}_counter_collection.csv; }
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_WAVES
