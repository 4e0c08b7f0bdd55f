"""bench.py end to end on the GPU box: the one JSON line carries what the contract asks for, on the default
config and through the self-launched N = 2 path (two ranks sharing the one GPU over gloo: RCCL refuses two
ranks on one device, the collective and sharding code is the same)."""

import json
import os
import subprocess
import sys

import pytest

import conftest

pytestmark = pytest.mark.gpu


def _bench(env, *argv, timeout=500):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py")] + list(argv), env=e,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_two_self_launched_ranks(device):
    out = _bench({"BLGNN_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--config", "1", "--steps", "3", "--warmup", "1",
                 "--no-extra", "--no-cpu-baseline", "--min-seconds", "0.05")
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["pairs_per_step"] == 2 * out["config"]["pairs_per_step_per_gpu"]
    assert out["config"]["drugs"] == 2 * 1326                     # weak scaling: the molecule set grows with N


def test_bench_default_line_has_the_contract_fields(device):
    out = _bench({}, "--steps", "2", "--warmup", "1", "--no-extra", "--cpu-seconds", "2", "--min-seconds", "0.05")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["config"]["baseline_config_index"] == 2 and "workload" in out["config"]
    r = out["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["launches"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["cores"] >= 1
