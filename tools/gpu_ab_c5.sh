# bench.py on both workloads for each environment setting: bash tools/gpu_ab_c5.sh "ENV=.." ...
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "$@"; do
  for wl in c3 c5; do
    env $cfg python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab5.log 2> gpurun_out/ab5.err || { tail -20 gpurun_out/ab5.err; exit 1; }
    python - "$cfg $wl" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/ab5.log').read().strip().splitlines()[-1])
r = d['roofline']
print(sys.argv[1], " value %.1f  ms/step %.1f  trace %.1f (ext %.1f mis %.1f shadow %.1f)  non-trace %.1f" % (d['value'], d['ms_per_step'], r['trace_ms_per_step'], r['trace_split_ms_per_step']['continuation'], r['trace_split_ms_per_step']['mis'], r['trace_split_ms_per_step']['shadow'], r['render_ms_per_step'] - r['trace_ms_per_step']))
PY
  done
done
