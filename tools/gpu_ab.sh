# bench.py once per environment setting (A/B of run-time knobs and AGPT_LIB_VARIANT builds): bash tools/gpu_ab.sh "ENV=.." ...
# AB_ARGS: extra bench.py arguments (default: the C3 headline workload), AB_STEPS: timed steps (default 3)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print("value %.1f Mrays/s (%.1f Msamples/s)  ms/step %.1f  trace %.1f (ext %.1f mis %.1f shadow %.1f)  non-trace %.1f  mean %s" % (
    d['value'], d['config']['msamples_per_s'], d['ms_per_step'], r['trace_ms_per_step'], r['trace_split_ms_per_step']['continuation'],
    r['trace_split_ms_per_step']['mis'], r['trace_split_ms_per_step']['shadow'], r["render_ms_per_step"] - r["trace_ms_per_step"], d["image_mean_rgb"]))
PY
}
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg python bench.py --steps ${AB_STEPS:-3} --warmup 1 --no-cpu-baseline --no-secondary $AB_ARGS > gpurun_out/ab.log 2> gpurun_out/ab.err || { tail -20 gpurun_out/ab.err; exit 1; }
  show gpurun_out/ab.log
  grep "trace clock\|trace stats\|shade clock" gpurun_out/ab.err | cut -c1-400 || true
done
