# Instruction-side SQ counters of one C3 bench step: bash tools/gpu_pmc_sq.sh  ->  gpurun_out/pmc_sq/ (two counters per pass,
# --kernel-trace only, every pass under its own timeout), then a per-kernel table
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sq
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; echo "pass $name: $@"; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/$name.log 2> $O/$name.err || { grep -m2 "error code\|Could not" $O/$name.err; echo "pass $name failed"; }; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_MISSES
run ic2 SQ_IFETCH SQ_WAIT_INST_ANY
run ic3 SQ_INSTS_VALU SQ_BUSY_CYCLES
run ic4 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY
run ic5 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run ic6 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
run ic7 SQ_WAVE_CYCLES SQ_WAIT_ANY
run ic8 SQC_DCACHE_REQ SQC_DCACHE_MISSES
run ic9 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
python3 - $O <<'PY'
import csv, glob, os, sys, collections
O = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(O, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({c for k in tab for c in tab[k]})
for k in tab:
    if "k_shade" in k or "k_trace_fast<0, 23, false, false" in k:
        print(k)
        for c in names:
            print("   %-28s %.6g" % (c, tab[k][c]))
PY
