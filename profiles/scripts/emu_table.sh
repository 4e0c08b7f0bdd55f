set -e
cd $GRAFT_REPO_ROOT
for n in 1 2 4 8; do
 for rep in 1 2; do
  extra=""; [ $n -gt 1 ] && extra="--emulate-world $n"
  timeout -k 10 300 python bench.py --config 3 $extra --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/emu_c3_n$n.json 2> gpurun_out/emu_c3_n$n.err || { tail -20 gpurun_out/emu_c3_n$n.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/emu_c3_n$n.json") if l.startswith("{")][0])
print("configs[3] N=$n", round(d["ms_per_step"],3), "min", round(d["timing"]["ms_per_step_min"],3) if "timing" in d else "", "max", round(d["timing"]["ms_per_step_max"],3) if "timing" in d else "", "interaction", round(d["config"]["interaction_level_ms"],3), d["config"].get("interaction_level","")[:22])
PY
 done
done
for n in 8; do
  timeout -k 10 300 python bench.py --config 1 --emulate-world $n --no-extra --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/emu_c1_n$n.json 2> gpurun_out/emu_c1_n$n.err
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/emu_c1_n$n.json") if l.startswith("{")][0])
print("configs[1] N=$n", round(d["ms_per_step"],4), "interaction", round(d["config"]["interaction_level_ms"],4), d["config"].get("interaction_level","")[:22])
PY
done
