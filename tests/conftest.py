import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bilevel-graph-neural-network_amd"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return "cuda:0"


def pytest_sessionfinish(session, exitstatus):
    """Write the measured parity errors of this session (tests/helpers.py:ERROR_LOG) where the GPU box's
    output directory is merged back from: gpurun_out/parity_errors.json."""
    try:
        import json
        import helpers
        if not helpers.ERROR_LOG:
            return
        summary = {}
        for test, rows in helpers.ERROR_LOG.items():
            summary[test] = {"comparisons": len(rows), "max_abs": max(r["abs"] for r in rows),
                             "max_scaled": max(r["scaled"] for r in rows),
                             "max_ref_magnitude": max(r["ref_max"] for r in rows)}
            c32 = [r for r in rows if "cpu32_scaled" in r]
            if c32:                                          # helpers.compare3: the fp32 CPU oracle's own error beside the GPU's
                summary[test].update({"cpu32_max_abs": max(r["cpu32_abs"] for r in c32),
                                      "cpu32_max_scaled": max(r["cpu32_scaled"] for r in c32),
                                      "gpu_max_scaled_where_cpu32_known": max(r["scaled"] for r in c32),
                                      "worst_gpu_over_cpu32": max(r["scaled"] / max(r["cpu32_scaled"], 1e-12) for r in c32)})
        out = {"bar": helpers.TOL, "worst_abs": max(v["max_abs"] for v in summary.values()),
               "worst_scaled": max(v["max_scaled"] for v in summary.values()), "tests": summary}
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        import torch
        name = "parity_errors.json" if torch.cuda.is_available() else "parity_errors_cpu.json"
        with open(os.path.join(d, name), "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
    except Exception:                                        # reporting must never fail a run
        pass


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    try:
        import helpers
        rows = [r for v in helpers.ERROR_LOG.values() for r in v]
        if rows:
            terminalreporter.write_line(
                "parity: %d comparisons, worst abs %.3e, worst scaled %.3e (bar %.0e); per test -> "
                "gpurun_out/parity_errors.json" % (len(rows), max(r["abs"] for r in rows),
                                                    max(r["scaled"] for r in rows), helpers.TOL))
    except Exception:
        pass
