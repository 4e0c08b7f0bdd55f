set -e
cd $GRAFT_REPO_ROOT
for n in 1 2; do for ser in 0 1; do
  extra=""; [ $n -gt 1 ] && extra="--emulate-world $n"
  BLGNN_ENGINE_SERIALIZE=$ser timeout -k 10 300 python bench.py --config 3 $extra --no-extra --no-cpu-baseline --steps 10 --warmup 3 --min-seconds 3 > gpurun_out/ser_n${n}_$ser.json 2> gpurun_out/ser.err || { tail -20 gpurun_out/ser.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ser_n${n}_$ser.json") if l.startswith("{")][0])
print("N=$n serialize=$ser", round(d["ms_per_step"],3), d["timing"]["blocks_of_K_steps"], round(d["timing"]["ms_per_step_min"],3), round(d["timing"]["ms_per_step_max"],3), d["config"]["two_stream_overlap"])
PY
done; done
