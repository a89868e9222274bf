set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/tests_q.log 2>&1 || { tail -40 gpurun_out/tests_q.log; exit 1; }
tail -2 gpurun_out/tests_q.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_q.log 2> gpurun_out/bench_q.err || { tail -30 gpurun_out/bench_q.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_q.log').read().strip().splitlines()[-1])
r = d['roofline']
print("value %.1f Mrays/s  ms/step %.1f  trace %.1f (ext %.1f mis %.1f shadow %.1f)  non-trace %.1f  frac %.3f" % (
    d['value'], d['ms_per_step'], r['trace_ms_per_step'], r['trace_split_ms_per_step']['continuation'], r['trace_split_ms_per_step']['mis'],
    r['trace_split_ms_per_step']['shadow'], r['render_ms_per_step'] - r['trace_ms_per_step'], r['frac']))
PY
