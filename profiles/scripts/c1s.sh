set -e
cd $GRAFT_REPO_ROOT
for n in 4 8; do for f in 0 1; do
  BLGNN_BENCH_SHARD_UPPER=$f timeout -k 10 300 python bench.py --config 1 --emulate-world $n --no-extra --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/c1s.json 2> gpurun_out/c1s.err || { tail -20 gpurun_out/c1s.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/c1s.json") if l.startswith("{")][0])
print("configs[1] N=$n shard=$f", round(d["ms_per_step"],4), "pass", round(d["config"]["interaction_level_ms"],4), d["config"].get("interaction_level","")[:12])
PY
done; done
BLGNN_BENCH_SHARD_UPPER=1 timeout -k 10 300 python bench.py --config 1 --rccl-one-rank --no-extra --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/c1r.json 2> gpurun_out/c1r.err
timeout -k 10 300 python bench.py --config 1 --no-extra --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/c1p.json 2> gpurun_out/c1p.err
python - <<PY
import json
for f in ("c1p","c1r"):
    d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][0])
    print(f, round(d["ms_per_step"],4), "pass", round(d["config"]["interaction_level_ms"],4), d["config"].get("interaction_level","")[:12])
PY
