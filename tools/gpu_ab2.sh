set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for args in "$@"; do
  echo "== $args"
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline $args > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
  python - <<'PY'
import json
d = json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1])
r = d['roofline']
print("value %.1f Mrays/s  ms/step %.1f  trace %.1f  non-trace %.1f launches %d" % (d['value'], d['ms_per_step'], r['trace_ms_per_step'], r['render_ms_per_step'] - r['trace_ms_per_step'], r['launches']))
PY
done
