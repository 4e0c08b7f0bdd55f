set -e
cd $GRAFT_REPO_ROOT
for extra in "" "--rccl-one-rank" "--rccl-one-rank --no-shard-upper"; do
  tag=$(echo "$extra" | tr -d ' ')
  timeout -k 10 300 python bench.py --config 3 --no-extra --no-cpu-baseline --steps 10 --warmup 3 --min-seconds 2 $extra > gpurun_out/rccl1$tag.json 2> gpurun_out/rccl1$tag.err || { tail -30 gpurun_out/rccl1$tag.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/rccl1$tag.json") if l.startswith("{")][0])
print("[$extra]", round(d["ms_per_step"],3), round(d["timing"]["ms_per_step_min"],3), round(d["timing"]["ms_per_step_max"],3), "pass", round(d["config"]["interaction_level_ms"],3), d["config"]["exchange"][:40], "|", d["config"].get("interaction_level","")[:24])
PY
done
