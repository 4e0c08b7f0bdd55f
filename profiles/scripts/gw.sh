set -e
cd $GRAFT_REPO_ROOT
for cap in 2048 4096 8192 32768 1024; do
  BLGNN_GATHER_WAVE_BLOCKS=$cap timeout -k 10 300 python bench.py --config 3 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/gw_$cap.json 2> gpurun_out/gw.err || { tail -30 gpurun_out/gw.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/gw_$cap.json") if l.startswith("{")][0])
g=d["interaction_gather"]
print("cap $cap: gather avg_us", round(g["avg_us"],1), "gathered GB/s", round(g["gathered_GBps"]), "step", round(d["ms_per_step"],3), "pass", round(d["config"]["interaction_level_ms"],3))
PY
done
