set -e
cd $GRAFT_REPO_ROOT
for hd in 128 32 64 256 512; do
  BLGNN_GATHER_HUB_BLOCKS=$hd timeout -k 10 300 python bench.py --config 3 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/hub1.json 2> gpurun_out/hub.err || { tail -30 gpurun_out/hub.err; exit 1; }
  BLGNN_GATHER_HUB_BLOCKS=$hd timeout -k 10 300 python bench.py --config 3 --emulate-world 8 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/hub8.json 2> gpurun_out/hub.err || { tail -30 gpurun_out/hub.err; exit 1; }
  python - <<PY
import json
a=json.loads([l for l in open("gpurun_out/hub1.json") if l.startswith("{")][0])
b=json.loads([l for l in open("gpurun_out/hub8.json") if l.startswith("{")][0])
print("hub blocks $hd: N=1 gather us", round(a["interaction_gather"]["avg_us"],1), "step", round(a["ms_per_step"],3), "| N=8 pass", round(b["config"]["interaction_level_ms"],3), "step", round(b["ms_per_step"],3))
PY
done
