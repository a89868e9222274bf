set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1 || true
grep -c . gpurun_out/pmc/counters_list.txt
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmc -o p1 -- python3 bench.py --steps 1 --warmup 0 --spp 4 --no-cpu-baseline > gpurun_out/pmc/p1.log 2> gpurun_out/pmc/p1.err || { tail -20 gpurun_out/pmc/p1.err; exit 1; }
ls gpurun_out/pmc
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('gpurun_out/pmc/p1_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][:40]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    if 'k_trace' in k or 'k_shade' in k:
        print(k)
        for c, v in sorted(d.items()): print('   %-26s %.4g' % (c, v))
        if d.get('SQ_ACTIVE_INST_VALU'):
            print('   lane util (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64)) = %.3f' % (d['SQ_THREAD_CYCLES_VALU'] / (d['SQ_ACTIVE_INST_VALU'] * 64)))
        if d.get('SQ_WAVE_CYCLES'):
            print('   wait_any / wave_cycles = %.3f ; active_valu/wave_cycles = %.3f' % (d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES'], d['SQ_ACTIVE_INST_VALU'] / d['SQ_WAVE_CYCLES']))
        if d.get('SQ_WAVES'):
            print('   valu insts / wave = %.1f' % (d['SQ_INSTS_VALU'] / d['SQ_WAVES']))
PY
