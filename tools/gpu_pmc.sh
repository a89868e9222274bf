# PMC passes for bench.py's roofline: bash tools/gpu_pmc.sh <workload c3|c5> <tag>  ->  gpurun_out/pmc_<tag>/ + pmc_<tag>.json
# (copy the json and the *_kernel_stats.csv into profiles/).  One counter group per pass, --kernel-trace only.
set -e
cd $GRAFT_REPO_ROOT
WL=${1:-c3}; TAG=${2:-r03_$WL}
O=gpurun_out/pmc_$TAG
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; echo "pmc pass $name ..."; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O -o $name -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/$name.log 2> $O/$name.err || { tail -20 $O/$name.err; exit 1; }; echo "pmc pass $name done"; }
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY
run rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
run wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
echo "kernel stats ..."
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > $O/stats.log 2> $O/stats.err || { tail -20 $O/stats.err; exit 1; }
KEY=$(python3 - <<PY
import json
d = json.loads(open("$O/sq.log").read().strip().splitlines()[-1])
print("$WL:%dx%d:%dspp:world1:scale1" % (d["config"]["resolution"][0], d["config"]["resolution"][1], d["config"]["spp"]))
PY
)
python3 tools/pmc_json.py $O "$KEY" gpurun_out/pmc_$TAG.json | cut -c1-1500
head -6 $O/stats_kernel_stats.csv | cut -c1-60,200-330
