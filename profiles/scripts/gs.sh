set -e
cd $GRAFT_REPO_ROOT
for n in 1 2 3 4 2 1; do
  timeout -k 10 300 python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --min-seconds 2 --gmn-streams $n > gpurun_out/gs_$n.json 2> gpurun_out/gs.err || { tail -30 gpurun_out/gs.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/gs_$n.json") if l.startswith("{")][0])
print("streams $n: step", round(d["ms_per_step"],3), round(d["timing"]["ms_per_step_min"],3), round(d["timing"]["ms_per_step_max"],3), "frac", round(d["roofline"]["frac"],4) if d.get("roofline") else None)
PY
done
