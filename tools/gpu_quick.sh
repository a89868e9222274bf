# GPU parity suite + a short bench line + optional env A/B settings: bash tools/gpu_quick.sh ["ENV=.. ENV2=.." ...]
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/tests_q.log 2>&1 || { tail -40 gpurun_out/tests_q.log; exit 1; }
tail -2 gpurun_out/tests_q.log
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print("value %.1f Mrays/s (queries %.1f, %.1f Msamples/s)  ms/step %.1f  trace %.1f (ext %.1f mis %.1f shadow %.1f)  non-trace %.1f" % (
    d['value'], d['config']['reference_queries_mrays_s'], d['config']['msamples_per_s'], d['ms_per_step'], r['trace_ms_per_step'],
    r['trace_split_ms_per_step']['continuation'], r['trace_split_ms_per_step']['mis'],
    r['trace_split_ms_per_step']['shadow'], r['render_ms_per_step'] - r['trace_ms_per_step']))
PY
}
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_q.log 2> gpurun_out/bench_q.err || { tail -30 gpurun_out/bench_q.err; exit 1; }
show gpurun_out/bench_q.log
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
  show gpurun_out/ab.log
  grep "trace clock\|trace stats" gpurun_out/ab.err | cut -c1-330 || true
done
