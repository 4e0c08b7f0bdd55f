"""host-side timing of the engine's step pieces at an emulated rank count (diagnosis of the N = 2 stalls)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--config", "3", "--emulate-world", os.environ.get("EMU", "2"), "--no-extra", "--no-cpu-baseline",
            "--steps", "10", "--warmup", "3", "--min-seconds", "0.5"] + sys.argv[1:]
import bench
import torch
from blgnn import engine as E

log = []
def timed(cls, name):
    fn = getattr(cls, name)
    def w(self, *a, **k):
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        log.append((name, (time.perf_counter() - t0) * 1e3))
        return r
    setattr(cls, name, w)
for n in ("step", "_feed", "exchange", "_replay_upper", "synchronize", "rows_of", "_weights_stamp"):
    timed(E.BiLevelEngine, n)
orig_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    t0 = time.perf_counter()
    orig_replay(self)
    log.append(("graph.replay", (time.perf_counter() - t0) * 1e3))
torch.cuda.CUDAGraph.replay = replay
orig_es = torch.cuda.Event.synchronize
def es(self):
    t0 = time.perf_counter()
    orig_es(self)
    log.append(("event.sync", (time.perf_counter() - t0) * 1e3))
torch.cuda.Event.synchronize = es
orig_copy = torch.Tensor.copy_
def cp(self, *a, **k):
    t0 = time.perf_counter()
    r = orig_copy(self, *a, **k)
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 2.0:
        log.append(("copy_ %s<-%s %s" % (self.device.type, a[0].device.type, tuple(self.shape)), dt))
    return r
torch.Tensor.copy_ = cp
args = bench.parse()
D = bench.Dist(args)
res = bench.run_bilevel(3, args, D, "f16x3", args.steps, args.warmup, args.min_seconds, want_roofline=False)
print("ms/step", res["ms_per_step"], res["ms_per_step_min"], res["ms_per_step_max"])
slow = [(n, t) for n, t in log if t > 5.0]
print("calls > 5 ms:", len(slow), "of", len(log))
from collections import Counter
print(Counter(n for n, _ in slow))
for n, t in slow[:40]:
    print("  %-16s %8.2f ms" % (n, t))
