#!/usr/bin/env python3
"""Headline benchmark: DDI edge-pairs/sec of the bi-level forward on N MI355X.

``--config`` selects one of BASELINE.json's five configs (0-based); the default is configs[2], the
largest single-GPU configuration:

  0  DrugBank shape, lower GIN x5 + upper GAT x3, d=64,  32 pairs per GPU per step        (weak)
  1  DrugBank shape, lower GIN x5 + upper GAT x3, d=128, 4 096 pairs per GPU per step     (weak)
  2  DrugCombo shape, multi-label head, lower GIN + GMN cross-graph attention x3, d=256,
     8 192 pairs per GPU per step in pair mode (16 384 molecule instances, ~480 k atoms)   (weak)
  3  100 k drugs x ~40 atoms, lower GIN x5 (d=256, multi-scale readout 1 280) sharded over the ranks
     -> all-gather of the pooled rows -> upper GAT x3 on 1 M interaction pairs; 65 536 pairs scored
     per step, split over the ranks                                                         (strong)
  4  d=512, GMN on, pairs drawn from 1 M interaction edges, 8 192 pairs per step split over
     the ranks                                                                              (strong)

One step = one complete forward of the reference's evaluation path (src/train.py:75-108,185-220) on
one batch of edge pairs.  Bi-level configs: lower sweep over this rank's molecules -> readout ->
[all-gather of pooled drug embeddings over RCCL when N > 1] -> interaction-level conv x3 -> pair
scores; every step scores a FRESH random batch (one 16-byte-per-pair H2D id copy per step, inside the
timed region).  GMN configs: NodeEmbedding / GMNPropagator x3 over the merged pair batch -> readout
-> pair scores, cycling through pre-built batches resident in HBM.

``--gpus N`` without a torchrun environment starts the N ranks itself (child processes, RCCL).
Timing: W warm-up steps, then blocks of EXACTLY K steps bracketed by barrier + synchronize, MAX over
ranks; blocks repeat until >= 0.5 s has been timed and the MEDIAN block is reported.
Prints ONE JSON line on rank 0 with `roofline` for the config's dominant kernel (HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample).
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "bilevel-graph-neural-network_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

SEED = 20200614
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA (same table)
PEAK_HBM_GBPS = 8000.0
MIN_TIMED_SECONDS = 0.5

CONFIGS = {
    0: dict(kind="bilevel", shape="drugbank", drugs=1326, pairs=29240, d=64, batch=32, scaling="weak",
            name="DrugBank-shape DDI, lower GIN x5 + upper GAT x3, d=64, batch=32 edge pairs per GPU "
                 "(BASELINE configs[0])"),
    1: dict(kind="bilevel", shape="drugbank", drugs=1326, pairs=29240, d=128, batch=4096, scaling="weak",
            name="DrugBank-shape DDI, lower GIN x5 + upper GAT x3, d=128, batch=4096 edge pairs per GPU "
                 "(BASELINE configs[1])"),
    2: dict(kind="gmn", shape="drugcombo", drugs=3376, pairs=50263, d=256, batch=8192, scaling="weak",
            dataset="drugcombo", edge_types=2,
            name="DrugCombo-shape multi-label DDI, lower GIN + GMN cross-graph attention x3, d=256, "
                 "batch=8192 edge pairs per GPU, pair mode (BASELINE configs[2])"),
    3: dict(kind="bilevel", shape="atoms40", drugs=100000, pairs=1000000, d=256, batch=65536, scaling="strong",
            name="synthetic bi-level: 100k drugs x ~40 atoms, lower GIN x5 d=256 sharded + all-gather + upper "
                 "GAT x3 on 1M pairs, 65536 edge pairs per step over all GPUs (BASELINE configs[3])"),
    4: dict(kind="gmn", shape="atoms40", drugs=100000, pairs=1000000, d=512, batch=8192, scaling="strong",
            dataset="drugbank", edge_types=1,
            name="synthetic 1M interaction edges, lower GIN + GMN x3, d=512, 8192 edge pairs per step over all "
                 "GPUs, pair mode (BASELINE configs[4])"),
    # not a BASELINE config: the DrugCombo interaction level (MetaLayerWrapper + NodeModelAggrByEdge('gat'), one
    # GATConv per edge type summed) at the real data's size -- 3 376 molecules, 50 263 labelled pairs over two edge types --
    # so that every hot-path row has a number (``extra.drugcombo_interaction_level``)
    5: dict(kind="bilevel", shape="drugcombo", drugs=3376, pairs=50263, d=128, batch=4096, scaling="weak",
            dataset="drugcombo", edge_types=2,
            name="DrugCombo-shape bi-level: lower GIN x5 + MetaLayer (GAT per edge type, summed) x3 + multi-label head, "
                 "d=128, batch=4096 edge pairs (interaction level of SURVEY 8a10; not a BASELINE config)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="index into BASELINE.json configs (default 2: DrugCombo d=256 B=8192 GMN)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extras: configs[1] summary and the d=256 SpMM/readout micro-roofline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-shard-upper", action="store_true",
                    help="N > 1, >= 32 768 interaction rows: keep the interaction level replicated on every rank instead of "
                         "target-row sharded (blgnn/engine.py shard_upper)")
    ap.add_argument("--rccl-one-rank", action="store_true",
                    help="N = 1, bi-level configs: initialise RCCL with one rank and run the engine's all-gathers (the sweep's, "
                         "and with >= 32 768 interaction rows the sharded interaction pass's) as real RCCL calls between "
                         "the captured graphs -- what the collectives cost a step apart from the link transfer")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="profiling aid: build rank 0's share of an N-GPU run on one GPU (no collective)")
    ap.add_argument("--fuse-lower", action="store_true",
                    help="run the lower level as ONE launch (blgnn_gin_stack[_h3]) instead of one per GIN layer "
                         "(default wherever the f16x3 stack applies: d=128)")
    ap.add_argument("--no-fuse-lower", action="store_true", help="one launch per GIN layer + a readout launch")
    ap.add_argument("--transform", choices=("f16x3", "bf16x6", "f32"), default="f16x3",
                    help="how the dense transforms run: fp16 two-piece split (default), bf16 three-piece split, "
                         "or exact-f32 MFMA -- all fp32-equivalent")
    ap.add_argument("--exact-f32", action="store_true", help="same as --transform f32")
    ap.add_argument("--lower-cus", type=int, default=-1,
                    help="bi-level configs: compute units per XCD (of 32) reserved for the molecular sweep's stream; "
                         "the interaction-level stream gets the rest.  0 = ordinary streams; -1 = the measured "
                         "optimum (20 at 1 GPU, 16 at 2, none from 4 GPUs on / for the large configs)")
    ap.add_argument("--train", action="store_true",
                    help="instead of the forward metric: time TRAINING steps of the bi-level model at the DrugBank shape "
                         "(molecular sweep with gradients -> interaction level -> BCE -> backward -> SGD; "
                         "src/train.py:75-108,177-182) and print one JSON line for them; with --config 5 the DrugCombo "
                         "bi-level model, with --config 2 / 4 the pair-batched GMN model (eager)")
    ap.add_argument("--readout-with-upper", action="store_true")
    ap.add_argument("--gmn-fused-amax", action="store_true",
                    help="GMN configs: the producers of the f_node input merge its row magnitudes (atomic maximum) instead "
                         "of a pass over it (measured: no gain)")
    ap.add_argument("--no-prune-last", action="store_true",
                    help="sharded bi-level configs: compute every row of the last interaction layer, not only the rows the "
                         "rank's pairs read")
    ap.add_argument("--no-shared-prefix", action="store_true",
                    help="GMN configs: compute every copy of a molecule in the pair batch (the reference's batch as is) "
                         "instead of running the layers in front of the first cross-graph attention on the distinct "
                         "molecules")
    ap.add_argument("--no-gin-sp", action="store_true",
                    help="wide GIN layers as one fused launch instead of aggregation + two pre-split GEMMs")
    ap.add_argument("--no-pooled-tail", action="store_true",
                    help="GMN configs: write every propagator's node rows and run the readout as its own launch instead of "
                         "pooling in the GEMM epilogues (and the last second Linear per graph)")
    ap.add_argument("--train-eager", action="store_true",
                    help="--train: run the iteration launch by launch instead of replaying its HIP graph")
    ap.add_argument("--no-gmn-sp", action="store_true",
                    help="GMN configs: the in-kernel operand split of blgnn_linear_h3 instead of the pre-split pipeline")
    ap.add_argument("--gmn-match-fork", choices=("message", "edge", "split"), default=None,
                    help="GMN configs: the attention's side stream forks beside the message GEMM or beside the edge kernel")
    ap.add_argument("--gmn-match-streams", type=int, default=1, choices=(1, 2),
                    help="GMN configs: side streams of the cross-graph attention's size-class kernels")
    ap.add_argument("--gmn-streams", type=int, default=1,
                    help="GMN configs: consecutive steps (independent pair batches) alternate over this many streams, i.e. "
                         "that many forwards are in flight (2: 14.35 -> 14.04 ms/step at configs[2], checked against the same "
                         "batch run alone; default 1 = one forward at a time, the regime the roofline pass times kernels in)")
    ap.add_argument("--no-fuse-readout", action="store_true",
                    help="bi-level configs: a separate segment-reduce launch instead of the readout fused into the "
                         "GIN layers' epilogues")
    ap.add_argument("--no-overlap", action="store_true",
                    help="replay the lower and upper graphs of a step back to back on one stream")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying captured HIP graphs")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_SECONDS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# --gpus N without a launcher: start the N ranks here, before anything touches the GPU
# ---------------------------------------------------------------------------------------------------
def self_launch(args):
    """Spawn ``args.gpus`` child processes of this script (one per GPU, RCCL rendezvous on 127.0.0.1),
    pass rank 0's stdout through and exit non-zero if any rank fails.  The parent never initialises
    the GPU (and never re-execs)."""
    n = args.gpus
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "BLGNN_BENCH_CHILD": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = None if r == 0 else subprocess.DEVNULL          # rank 0 prints the one JSON line
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rcs = [None] * n
    deadline = time.time() + 3300
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed or time.time() > deadline:
            for p in procs:                                   # exact PIDs we started
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            why = "rank(s) %s exited non-zero" % failed if failed else "timed out"
            raise SystemExit("bench.py --gpus %d: %s (%s): no result line is valid" % (n, why, rcs))
        time.sleep(0.2)
    return 0


# ---------------------------------------------------------------------------------------------------
# timing helpers
# ---------------------------------------------------------------------------------------------------
class KernelTimer(object):
    """HIP-event timing of one op's launches on the stream they are launched on
    (torch's current stream == the stream handed to libblgnn)."""

    def __init__(self):
        self.pairs, self.work, self.names, self.enabled = [], [], [], False

    def wrap(self, name, fn, work_of, select):
        import torch

        def wrapped(*a, **k):
            if not self.enabled or not select(*a, **k):
                return fn(*a, **k)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            self.work.append(work_of(*a, **k))
            self.names.append(name)
            return out
        return wrapped

    def result(self):
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        return ms, self.work, self.names


def timed_blocks(run_block, min_seconds, max_blocks=200):
    """run_block() -> seconds for exactly K steps (barrier + synchronize on both sides, MAX over ranks, so
    every rank sees the same number and takes the same decision).  Repeats until ``min_seconds`` have been
    timed; returns (median, all block times)."""
    times = []
    while True:
        times.append(run_block())
        if sum(times) >= min_seconds or len(times) >= max_blocks:
            break
    s = sorted(times)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2]), times


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota
    (a 1-GPU box exposes all 256 hardware threads but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def perturb_bn(model, seed):
    """eval BN, running stats perturbed by U(-0.1, 0.1)"""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(0.2 * (torch.rand(mod.running_mean.shape, generator=g) - 0.5))
                mod.running_var.copy_(1.0 + 0.2 * (torch.rand(mod.running_var.shape, generator=g) - 0.5))


def pmc_traffic(tag):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
    (profiles/summarize_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction), or None."""
    for rnd in ("r04", "r03", "r02", "r01"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", "%s_%s_pmc_summary.json" % (rnd, tag))))[
                "hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def roofline_of(kname, ms, work, mfma_mult, mfma_peak, traffic):
    """Two floors per launch -- algorithmic bytes at the HBM peak and EXECUTED matrix-core flops (3x / 6x /
    1x the algorithmic fp32 flops for f16x3 / bf16x6 / f32) at the dense peak of the MFMA dtype; the higher
    one is the binding roofline and is what `frac` is quoted against."""
    tot = sum(ms) * 1e-3
    flops, byts = sum(w[0] for w in work), sum(w[1] for w in work)
    tflops = flops / tot / 1e12 if tot else 0.0
    gbps = byts / tot / 1e9 if tot else 0.0
    exec_tflops = mfma_mult * tflops
    mfma_frac, hbm_frac = exec_tflops / mfma_peak, gbps / PEAK_HBM_GBPS
    by_hbm = hbm_frac >= mfma_frac
    n = max(len(ms), 1)
    return {"kernel": kname, "bound": "hbm" if by_hbm else "mfma",
            "achieved": gbps if by_hbm else exec_tflops,
            "peak": PEAK_HBM_GBPS if by_hbm else mfma_peak,
            "unit": "GB/s" if by_hbm else "TFLOP/s",
            "frac": hbm_frac if by_hbm else mfma_frac, "traffic": traffic,
            "alg_bytes_per_launch": byts / n, "alg_flops_per_launch": flops / n,
            "launches": len(ms), "avg_us": 1e3 * sum(ms) / n,
            "alg_hbm_GBps": gbps, "alg_hbm_frac": hbm_frac,
            "executed_mfma_tflops": exec_tflops, "executed_mfma_frac": mfma_frac, "mfma_peak_tflops": mfma_peak,
            "algorithmic_f32_tflops": tflops, "frac_of_f32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS}


MFMA_OF = {"f16x3": (3.0, PEAK_16BIT_MFMA_TFLOPS), "bf16x6": (6.0, PEAK_16BIT_MFMA_TFLOPS), "f32": (1.0, PEAK_F32_MFMA_TFLOPS)}
DTYPE_OF = {"f16x3": "f32 (dense transforms as f16x3 split MFMA, fp32-equivalent)",
            "bf16x6": "f32 (dense transforms as bf16x6 split MFMA, fp32-equivalent)", "f32": "f32"}


class Dist(object):
    """The process group as the bench needs it: barrier, MAX of a float, world/rank."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X; the hot path has no CPU fallback")
        # BLGNN_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
        self.backend = os.environ.get("BLGNN_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if self.backend == "nccl" and self.world > ndev:
            raise SystemExit("--gpus %d but only %d device(s) visible" % (self.world, ndev))
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        self.device = "cuda:%d" % dev_index
        self.dist = dist
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device(self.device))
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.one_rank_pg = False
        if self.world == 1 and getattr(args, "rccl_one_rank", False):
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            sock.close()
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(self.device))
            self.one_rank_pg = True
        if self.world > 1:
            joined = torch.ones(1, device=self.device if self.backend == "nccl" else "cpu")
            dist.all_reduce(joined)
            if int(joined.item()) != self.world:
                raise SystemExit("only %d of %d ranks joined" % (int(joined.item()), self.world))

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, v):
        import torch
        if self.world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            self.dist.barrier()
        if self.world > 1 or self.one_rank_pg:
            self.dist.destroy_process_group()


def block_runner(D, step_fn, sync_fn, steps):
    """-> run_block(): seconds for exactly ``steps`` steps, barrier + synchronize on both sides, MAX over ranks."""
    import torch
    counter = [0]

    def run_block():
        sync_fn()
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn(counter[0])
            counter[0] += 1
        sync_fn()
        D.barrier()
        torch.cuda.synchronize()
        return D.max(time.perf_counter() - t0)
    return run_block, counter


# ---------------------------------------------------------------------------------------------------
# bi-level configs (0, 1, 3): molecular sweep -> [all-gather] -> interaction level -> pair scores
# ---------------------------------------------------------------------------------------------------
def run_bilevel(cfg_id, args, D, transform, steps, warmup, min_seconds, want_roofline=True):
    import numpy as np
    import torch
    import blgnn
    from blgnn import ops
    from blgnn.batch import BatchData
    from blgnn.dist import ShardedLower
    from blgnn.engine import BiLevelEngine
    from blgnn.forward import all_molecule_pairs
    cfg = CONFIGS[cfg_id]
    world, rank, device = D.world, D.rank, D.device
    emu = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    eff_world = emu or world
    weak = cfg["scaling"] == "weak"
    d = cfg["d"]
    n_drugs = cfg["drugs"] * (eff_world if weak else 1)
    n_pairs = cfg["pairs"] * (eff_world if weak else 1)
    batch = cfg["batch"] if weak else cfg["batch"] // eff_world
    seed = SEED + cfg_id + 1
    edge_types = cfg.get("edge_types", 1)
    ds = blgnn.make_synthetic(n_drugs, n_pairs, shape=cfg["shape"], seed=seed, device=device, num_edge_types=edge_types)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset=cfg.get("dataset", "drugbank"),
                             lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=d, D_higher=d, device=device,
                             batch_size=batch, feature_transform=transform,
                             fuse_readout=not args.no_fuse_readout)
    flags.gin_sp = not getattr(args, "no_gin_sp", False)
    blgnn.set_flags(flags)
    torch.manual_seed(seed)
    data = blgnn.TrainData(ds, device=device, num_hyper_edge_feat=3 if edge_types > 1 else 0)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    model = blgnn.Model(data)
    perturb_bn(model, seed + 1)
    model.to(device).eval()
    sharded = ShardedLower(ds.molecules.gids, ds.molecules.sizes, 0 if emu else rank, eff_world)
    if emu:
        sharded_engine, engine_world = None, 1
    else:
        sharded_engine, engine_world = sharded, world

    # -- static inputs, resident in HBM before timing -------------------------------------
    lower_batch = BatchData(all_molecule_pairs(sharded.local_gids), ds, is_train=False, ignore_pairs=True)
    lower_batch.merge_data['merge'].csr()
    graph = ds.interaction_combo_nxgraph
    graph.init_x = torch.zeros((ds.num_graphs(), d * 5), dtype=torch.float32, device=device)
    ds.init = True
    rng = np.random.default_rng(seed + 17 + rank)
    template = BatchData(ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)], ds, is_train=False)
    Nn = int(lower_batch.merge_data['merge'].x.shape[0])
    nnz = int(lower_batch.merge_data['merge'].edge_index.shape[1])

    # dominant kernel: the fused GIN layer (CSR gather -> LDS -> 2 matrix-core GEMMs -> epilogue)
    timer = KernelTimer()
    saved = (ops.gin_layer, ops.gin_layer_x6, ops.gin_layer_h3)
    work_f32 = lambda csr, x, w1, b1, w2, *a, **kw: (
        2.0 * x.shape[0] * (w1.shape[0] * w1.shape[1] + w2.shape[0] * w2.shape[1]),
        4.0 * x.shape[0] * (x.shape[1] + w1.shape[1]) + 4.0 * csr.nnz + 4.0 * (csr.n + 1))
    work_split = lambda csr, x, w1p, b1, w2p, b2, N, *a, **kw: (
        2.0 * x.shape[0] * (x.shape[1] * N + N * N),
        4.0 * x.shape[0] * (x.shape[1] + N) + 4.0 * csr.nnz + 4.0 * (csr.n + 1))
    ops.gin_layer = timer.wrap("f32", saved[0], work_f32, lambda *a, **kw: True)
    ops.gin_layer_x6 = timer.wrap("bf16x6", saved[1], work_split, lambda *a, **kw: True)
    ops.gin_layer_h3 = timer.wrap("f16x3", saved[2], work_split, lambda *a, **kw: True)
    saved_stack = ops.gin_stack
    # the interaction-level gather (blgnn_gat_aggregate -> k_gather_wide on these degrees): bytes it gathers per launch =
    # one d-wide row per directed edge and self loop of the rows it computes
    gtimer = KernelTimer()
    saved_agg = ops.gat_aggregate

    def agg_work(csr, h, sc, stats, bias, *a, **kw):
        rows = kw.get("rows", a[8] if len(a) > 8 else None)      # (positional after bias: group .. accumulate, rows, packed)
        n, dd = h.shape
        frac = 1.0 if rows is None else float(rows.numel()) / max(n, 1)
        gathered = 4.0 * dd * (csr.nnz + n) * frac
        alg = 4.0 * dd * 2 * n + 4.0 * (csr.nnz + n) + 4.0 * (n + 1) + 16.0 * n          # GAT interaction layer
        return (0.0, alg, gathered)
    ops.gat_aggregate = gtimer.wrap("gat_aggregate", saved_agg, agg_work, lambda *a, **kw: True)

    def work_stack(csr, x0, tile_row, tile_mol, seg, layers, *a, **kw):
        # the one-launch lower level: L x the per-layer algorithmic work (SURVEY 8d defines the bytes per
        # layer as x read once + y written once + CSR; inside the stack the activations never leave LDS)
        rows, K0 = x0.shape
        N = layers[0]["N"] if "N" in layers[0] else layers[0]["w1"].shape[1]
        fl = by = 0.0
        for l in range(len(layers)):
            K = K0 if l == 0 else N
            fl += 2.0 * rows * (K * N + N * N)
            by += 4.0 * rows * (K + N) + 4.0 * csr.nnz + 4.0 * (csr.n + 1)
        return (fl, by)
    ops.gin_stack = timer.wrap("stack", saved_stack, work_stack, lambda *a, **kw: True)

    big = cfg["drugs"] > 10000
    lower_cus = ({1: 20, 2: 16}.get(eff_world, 0) if not big else 0) if args.lower_cus < 0 else args.lower_cus
    engine = BiLevelEngine(model, lower_batch, template, graph, sharded=sharded_engine, world_size=engine_world,
                           use_graph=not args.no_graph, overlap=not args.no_overlap,
                           fuse_lower=(False if args.no_fuse_lower else (True if args.fuse_lower else None)),
                           readout_with_upper=args.readout_with_upper,
                           lower_cus_per_xcd=lower_cus, exchange_transformed=(True if (emu or D.one_rank_pg) else None),
                           prune_last=(True if (emu and not args.no_prune_last) else (False if args.no_prune_last else None)),
                           shard_upper=(False if args.no_shard_upper else
                                        (True if ((emu or D.one_rank_pg) and (ds.num_graphs() >= 32768 or os.environ.get("BLGNN_BENCH_SHARD_UPPER") == "1"))
                                         else None)),
                           collective=(True if (D.one_rank_pg and not emu) else None))
    try:
        engine.capture()
        # a fresh random batch of pair ids every step (src/sampler.py:120-131), drawn ahead of the timed
        # region; the engine maps gid -> row and copies the 16 B/pair to the device inside the step
        pool = max(64, steps + warmup)
        fresh = [ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)] for _ in range(pool)]
        last = [None]

        def one(i):
            last[0] = engine.step(fresh[i % pool])
        for i in range(warmup):
            one(i)
        run_block, counter = block_runner(D, one, engine.synchronize, steps)
        counter[0] = warmup
        elapsed, blocks = timed_blocks(run_block, min_seconds)
        preds = last[0]
        assert preds.shape[0] == batch and bool(torch.isfinite(preds).all())
        # the interaction level + pair scorer alone: the captured upper graph replayed back to back (GPU time per pass)
        upper_ms = None
        if engine.use_graph and engine.g_upper[0] is not None:
            engine.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                for g_ in engine.g_upper[0]:           # (shard_upper: the pass's pieces, without the all-gathers between them)
                    g_.replay()
            e0.record()
            for _ in range(20):
                for g_ in engine.g_upper[0]:
                    g_.replay()
            e1.record()
            torch.cuda.synchronize()
            upper_ms = e0.elapsed_time(e1) / 20

        # HIP events cannot sit inside a replayed graph: the dominant kernel is timed over K steps launched
        # eagerly right after the timed region (same inputs, same stream).  A long matmul is queued ahead of
        # every step so that the step's launches wait in the queue and the events bracket back-to-back
        # execution, not host launch latency.
        roofline = None
        if want_roofline:                       # every rank: eager steps contain the all-gather
            for i in range(2):                  # untimed: the eager steps' activations come out of the allocator's pools
                engine.eager_step(fresh[i % pool])  # afterwards (a first eager step pays hipMalloc inside the events)
            torch.cuda.synchronize()
            timer.enabled = gtimer.enabled = rank == 0
            blk = torch.empty(4096, 8192, device=device)
            blk_w = torch.empty(8192, 8192, device=device)
            for i in range(min(steps, 20)):
                if not big:
                    torch.mm(blk, blk_w)
                else:
                    # launches of milliseconds: no queue blocker needed, but the next step's sweep must not start beside
                    # this step's interaction pass on the other stream (its events would time the two together)
                    torch.cuda.synchronize()
                engine.eager_step(fresh[i % pool])
            torch.cuda.synchronize()
            timer.enabled = gtimer.enabled = False
    finally:
        ops.gin_layer, ops.gin_layer_x6, ops.gin_layer_h3 = saved
        ops.gin_stack = saved_stack
        ops.gat_aggregate = saved_agg
    pairs_per_step = batch * world
    res = {"value": pairs_per_step / (elapsed / steps), "ms_per_step": 1e3 * elapsed / steps,
           "timed_blocks": len(blocks), "ms_per_step_min": 1e3 * min(blocks) / steps,
           "ms_per_step_max": 1e3 * max(blocks) / steps,
           "config": {"workload": cfg["name"], "drugs": n_drugs, "atoms_per_gpu": Nn, "mol_edges_per_gpu": nnz,
                      "interaction_pairs": int(len(ds.pair_gids)), "d": d, "pairs_per_step_per_gpu": batch,
                      "pairs_per_step": pairs_per_step, "fresh_batch_every_step": True,
                      "hip_graph": bool(engine.use_graph), "two_stream_overlap": bool(engine.overlap and not engine.serialized),
                      "readout_with_upper": bool(engine.readout_with_upper),
                      "fused_readout": bool(getattr(flags, "fuse_readout", False)) and engine.fused is None,
                      "lower_cus_per_xcd": engine.lower_cus_per_xcd, "fused_lower_stack": engine.fused is not None,
                      "exchange": ("first interaction layer's transform of the pooled rows [N_r, d+2]"
                                   if engine.exchange_transformed else "pooled rows [N_r, D_pool]"),
                      "interaction_level_ms": upper_ms,
                      "interaction_level": ("target-row sharded: %d pieces, an all-gather of the next layer's transformed rows "
                                            "between them" % engine.n_upper_segments) if engine.shard_upper else "replicated",
                      "parallelism": ("molecule shards x%d + all-gather(pooled) + pair shards" % world) if not engine.shard_upper
                      else ("molecule shards x%d + interaction target-row shards, %d all-gathers + pair shards"
                            % (world, engine.n_upper_segments))}}
    if want_roofline and rank == 0:
        ms, work, names = timer.result()
        if "stack" in names:
            # the dominant launch is the one-launch lower level
            sel = [i for i, n in enumerate(names) if n == "stack"]
            mult, peak = MFMA_OF[transform if transform != "bf16x6" else "f32"]
            kstack = ("k_gin_stack_h3 (whole lower level in one launch: %s GIN layers with CSR gather out of LDS, 2 GEMMs "
                      "each as 3 fp16 MFMA products per fp32 product, act/BN and the per-layer readout; algorithmic work "
                      "= L x the per-layer figure)" % flags.last_lower_lyr_num)
            res["roofline"] = roofline_of(kstack, [ms[i] for i in sel], [work[i] for i in sel], mult, peak,
                                          pmc_traffic("gin_stack_h3" + ("_cfg%d" % cfg_id if cfg_id != 1 else "")))
            names = []
        mode = max(set(names), key=names.count) if names else transform
        mult, peak = MFMA_OF[mode]
        kname = {"f16x3": "k_gin_layer_h3 (fused GIN layer: CSR gather in LDS + 2 GEMMs as 3 fp16 MFMA products per "
                          "fp32 product + act/BN)",
                 "bf16x6": "k_gin_layer_x6 (fused GIN layer: CSR gather in LDS + 2 GEMMs as 6 bf16 MFMA products per "
                           "fp32 product + act/BN)",
                 "f32": "k_gin_layer (fused GIN layer: CSR gather in LDS + 2 exact-f32 MFMA GEMMs + act/BN)"}[mode]
        tag = {"f16x3": "gin_layer_h3", "bf16x6": "gin_layer_x6", "f32": "gin_layer"}[mode]
        sel = [i for i, n in enumerate(names) if n == mode]
        if sel:
            sfx = "_cfg%d" % cfg_id if cfg_id != 1 else ""
            traffic = pmc_traffic(tag + sfx)
            if traffic is None:                          # (the d = 256 kernel's passes are filed under their width)
                traffic = pmc_traffic("%s_d%d%s" % (tag, d, sfx))
            res["roofline"] = roofline_of(kname, [ms[i] for i in sel], [work[i] for i in sel], mult, peak, traffic)
    if want_roofline and rank == 0:
        gms, gwork, _ = gtimer.result()
        if gms:
            tot = sum(gms) * 1e-3
            res["interaction_gather"] = {
                "kernel": "k_gather_wide (blgnn_gat_aggregate: per-source softmax weights + weighted neighbour-row sum of the "
                          "interaction-level GATConv, one workgroup per target row)",
                "launches": len(gms), "avg_us": 1e3 * sum(gms) / len(gms),
                "gathered_GBps": sum(w[2] for w in gwork) / tot / 1e9,
                "alg_hbm_GBps": sum(w[1] for w in gwork) / tot / 1e9,
                "alg_hbm_frac": sum(w[1] for w in gwork) / tot / 1e9 / PEAK_HBM_GBPS,
                "note": "the transformed table (N x d fp32) of the interaction graph is Infinity-Cache resident; the gathered "
                        "rate is what the micro-architecture guide's 7.4-8.6 TB/s for cache-resident row gathers compares with"}
    engine.close()
    res["_cpu"] = lambda seconds: cpu_baseline_bilevel(ds, flags, model, lower_batch, template, batch, seconds)
    return res


def cpu_baseline_bilevel(ds, flags, model, lower_batch, upper_batch, batch, seconds):
    """The CPU oracle (pure PyTorch restatement of the reference path) on the host cores, same workload, same
    weights; bounded sample.  Small molecule sets: whole steps.  Large ones (config 3): the lower sweep is
    timed on a 1/64 sample of this rank's molecules and scaled, the interaction level runs in full."""
    import numpy as np
    import torch
    import oracle
    import helpers
    from blgnn.batch import BatchData
    from blgnn.forward import all_molecule_pairs
    cores = usable_cores()
    torch.set_num_threads(cores)
    specs = helpers.layer_specs(flags)
    k = flags.last_lower_lyr_num
    sd = helpers.state_dict_cpu(model)
    of = helpers.oracle_flags(flags, ds)
    gids = lower_batch._graph_gids
    frac = 1
    if len(gids) > 8000:
        frac = 64
        sub = gids[:max(2, len(gids) // frac)]
        lower_batch = BatchData(all_molecule_pairs(sub), ds, is_train=False, ignore_pairs=True)
    pl = helpers.plain_batch(lower_batch, ds)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(upper_batch.batch_gids))
    ph0 = helpers.plain_batch(upper_batch, ds, with_pairs=ids)     # static graph layouts built once
    n, t_low, t_up = 0, 0.0, 0.0
    t0 = time.perf_counter()
    with torch.no_grad():
        while True:
            p = dict(pl)
            p["init_x"] = None
            a = time.perf_counter()
            oracle.run_layers(specs[:k + 1], sd, p, of)
            b = time.perf_counter()
            ph = dict(ph0)
            ph["init_x"] = p["init_x"]
            oracle.run_layers(specs[k + 1:], sd, ph, of, first=k + 1)
            c = time.perf_counter()
            t_low, t_up, n = t_low + (b - a), t_up + (c - b), n + 1
            if time.perf_counter() - t0 > seconds or n >= 200:
                break
    dt = frac * t_low + t_up
    what = "%d whole steps of the same workload" % n if frac == 1 else \
        "%d steps: lower sweep on 1/%d of the molecules (time scaled by %d) + full interaction level" % (n, frac, frac)
    return {"value": n * batch / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%s (oracle, torch CPU fp32, %d threads), %.1f s of CPU work"
                      % (what, cores, time.perf_counter() - t0)}


# ---------------------------------------------------------------------------------------------------
# GMN configs (2, 4): NodeEmbedding / GMNPropagator x3 over the merged pair batch -> readout -> scores
# ---------------------------------------------------------------------------------------------------
def run_gmn(cfg_id, args, D, transform, steps, warmup, min_seconds, want_roofline=True):
    import numpy as np
    import torch
    import blgnn
    from blgnn import ops
    from blgnn.batch import BatchData
    cfg = CONFIGS[cfg_id]
    world, rank, device = D.world, D.rank, D.device
    emu = args.emulate_world if (args.emulate_world > 1 and world == 1) else 0
    eff_world = emu or world
    weak = cfg["scaling"] == "weak"
    d = cfg["d"]
    batch = cfg["batch"] if weak else cfg["batch"] // eff_world
    seed = SEED + cfg_id + 1
    ds = blgnn.make_synthetic(cfg["drugs"], cfg["pairs"], shape=cfg["shape"], seed=seed, device=device,
                              num_edge_types=cfg["edge_types"])
    flags = blgnn.make_flags("lower_level_gmn", dataset=cfg["dataset"], lower_level_gnn_type="gin", D_lower=d,
                             device=device, batch_size=batch, feature_transform=transform)
    flags.gmn_fused_amax = bool(getattr(args, "gmn_fused_amax", False))
    flags.gmn_sp = not getattr(args, "no_gmn_sp", False)
    flags.gmn_shared_prefix = not getattr(args, "no_shared_prefix", False)
    flags.gmn_pooled_tail = not getattr(args, "no_pooled_tail", False)
    flags.gin_sp = not getattr(args, "no_gin_sp", False)
    flags.gmn_match_streams = int(getattr(args, "gmn_match_streams", 1))
    if getattr(args, "gmn_match_fork", None):
        flags.gmn_match_fork = args.gmn_match_fork
    blgnn.set_flags(flags)
    torch.manual_seed(seed)
    data = blgnn.TrainData(ds, device=device, num_hyper_edge_feat=3 if cfg["edge_types"] > 1 else 0)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    model = blgnn.Model(data)
    perturb_bn(model, seed + 1)
    model.to(device).eval()
    layers = list(model.layers[:-1])                       # ... LinkPred (no Loss)

    # pre-built pair batches, resident in HBM (merged graph, CSR, segment offsets); each rank draws its own
    rng = np.random.default_rng(seed + 17 + rank)
    nb = 4
    batches = []
    for _ in range(nb):
        b = BatchData(ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)], ds,
                      is_train=False, unique_graphs=False).prepare()   # CSR, segments, distinct molecules + row map, pair order
        batches.append(b)
    Nn = int(np.mean([b.merge_data['merge'].x.shape[0] for b in batches]))
    nnz = int(np.mean([b.merge_data['merge'].edge_index.shape[1] for b in batches]))

    # dominant kernel: the dense transforms of the propagator (f_message / f_node GEMMs, M = all atoms)
    timer = KernelTimer()
    saved = ops.linear
    min_rows = Nn // 16                                    # every GEMM over atoms (incl. the distinct-molecule ones), not the head's
    Nu = int(np.mean([b.shared_molecules().graph.num_nodes if b.shared_molecules() is not None else
                      b.merge_data['merge'].x.shape[0] for b in batches]))

    def lin_work(x, w_kn, *a, **kw):
        M, K = x.shape
        N = w_kn.shape[1]
        return (2.0 * M * N * K, 4.0 * M * (K + N) + 4.0 * K * N, (int(M), int(K), int(N)))
    ops.linear = timer.wrap(transform, saved, lin_work, lambda x, w_kn, *a, **kw: x.shape[0] >= min_rows)
    # the pre-split pipeline (default at these sizes): same products, operands in the sp format (4 B per element too)
    saved_sp = ops.linear_sp
    ops.linear_sp = timer.wrap("sp", saved_sp, lin_work, lambda x, w_kn, *a, **kw: x.shape[0] >= min_rows)
    import blgnn.layers_gmn as _lg
    import blgnn.layers_util as _lu
    assert _lg.ops is ops and _lu.ops is ops               # the modules call ops.linear through the module

    def forward(i):
        b = batches[i % nb]
        model.acts = [b.merge_data['merge'].x]
        with torch.no_grad():
            for layer in layers:
                model.acts.append(layer(model.acts[-1], b, model))
        return model.acts[-1]
    try:
        last = [None]

        n_streams = max(int(getattr(args, "gmn_streams", 1)), 1)
        lanes = [torch.cuda.Stream(device=device) for _ in range(n_streams)] if n_streams > 1 else None

        def one(i):
            if lanes is None:
                last[0] = forward(i)
            else:
                with torch.cuda.stream(lanes[i % n_streams]):
                    last[0] = forward(i)
        for i in range(nb):                                  # setup: every pre-built batch once (allocator pools, packed-weight
            one(i)                                           # caches), so a short --warmup does not time first-touch allocations
        for i in range(max(warmup, 2 * n_streams)):
            one(i)
        run_block, counter = block_runner(D, one, torch.cuda.synchronize, steps)
        counter[0] = warmup
        elapsed, blocks = timed_blocks(run_block, min_seconds)
        preds = last[0]
        assert preds.shape[0] == batch and bool(torch.isfinite(preds).all())
        if lanes is not None:
            # the pipelined steps compute what a step alone computes: the last timed step's batch again, alone on the default
            # stream (equal to the last-bit variation of the fused readouts' float atomics)
            alone = forward(counter[0] - 1)
            torch.cuda.synchronize()
            dev_ = float((alone - preds).abs().max()) / max(1.0, float(alone.abs().max()))
            assert dev_ < 1e-6, "pipelined forward differs from the same batch run alone: %.3e" % dev_
        shared_active = getattr(model.acts[1], "_blgnn_shared", None) is not None      # (too few repeats: not taken)
        pooled_active = getattr(model.acts[-3], "_blgnn_pooled", None) is not None     # (the last propagator's output)
        # the same steps on the plain path -- every copy of a molecule computed, node rows of every layer written and a
        # separate readout (disclosure next to the headline; not the reported value)
        other_ms = None
        if (flags.gmn_shared_prefix or flags.gmn_pooled_tail) and not args.no_extra:   # (same decision on every rank: barriers inside)
            from blgnn.config import FLAGS as live             # (set_flags copied ``flags`` into the process-wide bag)
            keep = (live.gmn_shared_prefix, live.gmn_pooled_tail)
            live.gmn_shared_prefix = live.gmn_pooled_tail = False
            try:
                for i in range(2):
                    one(i)
                counter[0] = warmup
                other_ms = 1e3 * run_block() / steps
            finally:
                live.gmn_shared_prefix, live.gmn_pooled_tail = keep
        # the host side of a batch (src/batch.py:105-144 + src/merged_graph.py: what the reference's evaluate loop does per
        # batch): the headline cycles through pre-built batches; here a FRESH pair batch is built for every step -- merge
        # of the molecule slices, CSR, segment offsets, distinct-molecule row map -- on a side stream while the previous
        # step runs, and the step's forward waits for it
        build_ms = incl_ms = None
        if not args.no_extra:
            side = torch.cuda.Stream(device=device)

            def build():
                with torch.cuda.stream(side):
                    b = BatchData(ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)], ds,
                                  is_train=False, unique_graphs=False).prepare()
                return b

            def forward_on(b):
                model.acts = [b.merge_data['merge'].x]
                with torch.no_grad():
                    for layer in layers:
                        model.acts.append(layer(model.acts[-1], b, model))
                return model.acts[-1]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                nxt = build()
                side.synchronize()
            build_ms = 1e3 * (time.perf_counter() - t0) / 3
            # A batch's tensors are allocated on the SIDE stream and read by the forward on the main stream: the caching
            # allocator would hand their memory to the next side-stream allocation the moment Python drops them, while
            # the forward that reads them may still be running.  A batch is therefore kept for two more steps, and
            # dropped only after the side stream has been told to wait for the end of the forward that used it.
            import collections
            keep = collections.deque()

            def pipelined_step():
                nonlocal_nxt = state["nxt"]
                main = torch.cuda.current_stream()
                main.wait_stream(side)
                forward_on(nonlocal_nxt)                             # enqueue the step ...
                ev = torch.cuda.Event()
                ev.record(main)
                keep.append((nonlocal_nxt, ev))
                if len(keep) > 2:
                    old, old_ev = keep.popleft()
                    side.wait_event(old_ev)
                    del old
                state["nxt"] = build()                               # ... and build the next batch under it
            state = {"nxt": nxt}
            for _ in range(3):                                       # warm the allocator pools of the rotating batches
                pipelined_step()
            torch.cuda.synchronize()
            D.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                pipelined_step()
            torch.cuda.synchronize()
            incl_ms = 1e3 * D.max(time.perf_counter() - t0) / steps
            keep.clear()
        if want_roofline and rank == 0:
            timer.enabled = True
            for i in range(min(steps, 8)):
                forward(i)
            torch.cuda.synchronize()
            timer.enabled = False
    finally:
        ops.linear = saved
        ops.linear_sp = saved_sp
    pairs_per_step = batch * world
    res = {"value": pairs_per_step / (elapsed / steps), "ms_per_step": 1e3 * elapsed / steps,
           "timed_blocks": len(blocks), "ms_per_step_min": 1e3 * min(blocks) / steps,
           "ms_per_step_max": 1e3 * max(blocks) / steps,
           "config": {"workload": cfg["name"], "drugs": cfg["drugs"], "atoms_per_gpu": Nn, "mol_edges_per_gpu": nnz,
                      "interaction_pairs": int(len(ds.pair_gids)), "d": d, "pairs_per_step_per_gpu": batch,
                      "pairs_per_step": pairs_per_step, "molecule_instances_per_gpu": 2 * batch,
                      "distinct_batches_cycled": nb, "output_logits": int(preds.shape[1]),
                      "parallelism": "pair shards x%d, no collective (pairs are independent)" % world}}
    res["config"]["shared_prefix"] = (
        {"on": True, "distinct_atoms_per_gpu": Nu,
         "what": "the first GIN layer and, of the first propagator, f_message, the message sum and the [x | S] columns "
                 "of f_node's first Linear run on the distinct molecules of the pair batch (the reference's batch holds "
                 "one copy per pair: identical rows); the attention, the u columns of that Linear (K-split, the "
                 "distinct-molecule part gathered in the epilogue) and every later layer run on all copies.  Same "
                 "outputs to fp32 rounding: tests/test_gpu_sp.py"} if shared_active else
        {"on": False, "why": "--no-shared-prefix" if not flags.gmn_shared_prefix else
         "%d of %d atoms are distinct: not worth a replication pass" % (Nu, Nn)})
    res["config"]["pooled_readout"] = (
        {"on": True,
         "what": "the mean readout is accumulated in the epilogue of the GEMM that produces the pooled rows; for the last "
                 "propagator that is f_node's FIRST Linear -- its second Linear has no activation behind it and commutes "
                 "with the mean, so it runs on one row per graph and the layer's node rows are never written.  Same "
                 "readout to fp32 rounding: tests/test_gpu_sp.py"} if pooled_active else
        {"on": False, "why": "--no-pooled-tail" if not flags.gmn_pooled_tail else "readout / path not eligible"})
    res["config"]["ms_per_step_plain_path"] = other_ms       # both restructurings off (FLAGS.gmn_shared_prefix / gmn_pooled_tail)
    # host cost of ONE fresh batch (merge + CSR + segments + distinct-molecule row map, device work included, synchronised) and
    # the step time with a fresh batch built for every step under the previous one
    res["config"]["batch_build_ms"] = build_ms
    res["config"]["ms_per_step_incl_batch_build"] = incl_ms
    res["value_incl_batch_build"] = (pairs_per_step / (incl_ms * 1e-3)) if incl_ms else None
    if want_roofline and rank == 0:
        ms, work, names = timer.result()
        mult, peak = MFMA_OF[transform]
        kname = {"f16x3": "k_linear_h3 (f_message / f_node GEMMs of the GMN propagator over all atoms: 3 fp16 MFMA "
                          "products per fp32 product, row-normalised operands, act/BN epilogue)",
                 "bf16x6": "k_linear_x6 (f_message / f_node GEMMs of the GMN propagator over all atoms: 6 bf16 MFMA "
                           "products per fp32 product, act/BN epilogue)",
                 "f32": "k_linear (f_message / f_node GEMMs of the GMN propagator, exact-f32 MFMA)"}[transform]
        tag = {"f16x3": "linear_h3", "bf16x6": "linear_x6", "f32": "linear"}[transform]
        if names and all(nm == "sp" for nm in names):
            kname = ("k_linear_sp (f_message / f_node GEMMs of the GMN propagator over all atoms on PRE-SPLIT operands: "
                     "3 fp16 MFMA products per fp32 product, 256 x 256 tiles, both operands by LDS-DMA, block-scaled "
                     "fp16 piece planes written by the producers' epilogues, act/BN epilogue)")
            tag = "linear_sp"
        res["roofline"] = roofline_of(kname, ms, work, mult, peak, pmc_traffic(tag + "_cfg%d" % cfg_id))
        # the same launches by GEMM shape (the step runs three to five): M x K x N, executed fp16 MFMA flops over the peak
        shapes = {}
        for t, w in zip(ms, work):
            M, K, N = w[2]
            e = shapes.setdefault(("all atoms" if M >= Nn // 2 else "distinct molecules", K, N), [0, 0.0, 0.0, 0])
            e[0] += 1; e[1] += t; e[2] += w[0]; e[3] += M
        res["roofline"]["by_shape"] = [
            {"rows": k[0], "M": v[3] // v[0], "K": k[1], "N": k[2], "launches": v[0], "avg_us": round(1e3 * v[1] / v[0], 1),
             "executed_mfma_frac": round(mult * v[2] / (v[1] * 1e-3) / 1e12 / peak, 4)}
            for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])]
    res["_cpu"] = lambda seconds: cpu_baseline_gmn(ds, flags, model, batch, seconds, seed)
    return res


def cpu_baseline_gmn(ds, flags, model, batch, seconds, seed):
    """The CPU oracle on a bounded sample of the same workload: pair batches of 256 pairs drawn like the
    GPU's (the forward is linear in the number of pairs: every pair is an independent component)."""
    import numpy as np
    import torch
    import oracle
    import helpers
    from blgnn.batch import BatchData
    cores = usable_cores()
    torch.set_num_threads(cores)
    specs = helpers.layer_specs(flags)
    sd = helpers.state_dict_cpu(model)
    of = helpers.oracle_flags(flags, ds)
    rng = np.random.default_rng(seed + 99)
    sub = min(256, batch)
    n = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        while True:
            b = BatchData(ds.pair_gids[rng.choice(len(ds.pair_gids), size=sub, replace=False)], ds,
                          is_train=False, unique_graphs=False)
            ids = torch.arange(2 * sub).view(sub, 2)
            oracle.run_layers(specs, sd, helpers.plain_batch(b, ds, with_pairs=ids), of)
            n += 1
            if time.perf_counter() - t0 > seconds or n >= 200:
                break
    dt = time.perf_counter() - t0
    return {"value": n * sub / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d forwards of %d-pair batches of the same workload (oracle, torch CPU fp32, %d threads; "
                      "includes its per-pair python loop for f_match as the reference has), %.1f s"
                      % (n, sub, cores, dt)}


# ---------------------------------------------------------------------------------------------------
# extras
# ---------------------------------------------------------------------------------------------------
def micro_roofline(device):
    """North-star kernel evidence: molecular SpMM (GIN aggregation) and segment-mean readout at
    d=256 on a 12 500-molecule x ~40-atom shard (BASELINE configs[3] per-GPU share), HIP-event
    timed; algorithmic bytes per SURVEY section 8(d)."""
    import torch
    import blgnn
    from blgnn import ops
    ds = blgnn.make_synthetic(12500, 10, shape="atoms40", seed=SEED + 102, device=device)
    m = ds.molecules.merge(ds.molecules.gids)["merge"]
    n, nnz, G, d = m.x.shape[0], m.edge_index.shape[1], m.num_graphs, 256
    x = torch.randn(n, d, device=device)
    csr, seg = m.csr(), m.segments()
    y = torch.empty_like(x)
    pooled = torch.empty((G, d), device=device)
    res = {"shape": {"atoms": n, "mol_edges": nnz, "molecules": G, "d": d}}

    def timeit(fn, reps=30):
        for _ in range(5):
            fn()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) for a, b in evs)
        return t[len(t) // 2] * 1e-3
    t = timeit(lambda: ops.spmm_sum(csr, x, 1.0, True, out=y))
    b = 4.0 * d * n * 2 + 4.0 * nnz + 4.0 * (n + 1)
    res["spmm_d256"] = {"bound": "hbm", "achieved": b / t / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": b / t / 1e9 / PEAK_HBM_GBPS, "alg_bytes": b, "us": t * 1e6,
                        "traffic": pmc_traffic("spmm_d256")}
    t = timeit(lambda: ops.segment_reduce(seg, x, "mean", out=pooled))
    b = 4.0 * d * n + 4.0 * d * G + 4.0 * (G + 1)
    res["readout_d256"] = {"bound": "hbm", "achieved": b / t / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                           "frac": b / t / 1e9 / PEAK_HBM_GBPS, "alg_bytes": b, "us": t * 1e6,
                           "traffic": pmc_traffic("readout_d256")}
    return res


def rendezvous_only(args):
    """Launcher rehearsal for boxes without N GPUs (tests/test_dist_cpu.py): the ranks meet over gloo,
    count themselves, rank 0 prints the result line's envelope with ``n_gpus`` = ranks that joined and
    ``value`` null.  Nothing is measured and no kernel runs."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    joined = torch.ones(1)
    if world > 1:
        dist.all_reduce(joined)
    if os.environ.get("BLGNN_BENCH_FAIL_RANK") == str(rank):
        raise SystemExit("rank %d: injected failure" % rank)
    if rank == 0:
        print(json.dumps({"metric": "DDI edge-pairs/sec (bi-level forward)", "value": None, "unit": "pairs/s",
                          "n_gpus": int(joined.item()), "steps": args.steps, "warmup": args.warmup,
                          "rendezvous_only": True, "scaling": CONFIGS[args.config]["scaling"],
                          "config": {"workload": CONFIGS[args.config]["name"]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def run_train(args, D, transform, steps, warmup):
    """Training step of the BI-GNN model (lower GIN x5 + upper GAT x3, d=128) at the DrugBank shape, one GPU:
    what src/train.py does per iteration -- the molecular sweep over ALL drugs with gradients, a sampled pair batch
    with one negative per positive, BCE, backward through both levels, optimizer step.  Every forward and backward
    step is a libblgnn kernel behind autograd nodes (blgnn/autograd.py); the optimizer is torch's."""
    import numpy as np
    import torch
    import blgnn
    import oracle
    import helpers
    from blgnn.batch import BatchData
    from blgnn.forward import all_molecule_pairs, get_initial_embd
    # --config 5: the DrugCombo bi-level model (MetaLayer interaction level, multi-label scorer, cross-entropy) under the
    # same captured iteration; any other --config: the DrugBank BI-GNN (configs[1] shape)
    combo = args.config == 5
    cfg = CONFIGS[5 if combo else 1]
    device = D.device
    d, batch = cfg["d"], 2048                                # 2 048 positives + 2 048 sampled negatives = 4 096 scored pairs
    ds = blgnn.make_synthetic(cfg["drugs"], cfg["pairs"], shape=cfg["shape"], seed=SEED + 7, device=device,
                              num_edge_types=cfg.get("edge_types", 1))
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset=cfg.get("dataset", "drugbank"),
                             lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=d, D_higher=d, device=device, batch_size=batch, feature_transform=transform)
    blgnn.set_flags(flags)
    torch.manual_seed(SEED)
    data = blgnn.TrainData(ds, device=device, num_hyper_edge_feat=3 if combo else 0)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    model = blgnn.Model(data).to(device)
    model.train()
    params = [p for k, p in model.named_parameters() if k.startswith("layers.")]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9)
    rng = np.random.default_rng(SEED)
    lower_cache = []
    losses = []
    captured = not getattr(args, "train_eager", False)

    def next_batch():
        # the host side of an iteration: draw positives, sample one negative per positive (src/batch.py:61-103)
        while True:
            pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)]
            bd = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=rng)
            if len(bd.batch_gids) == 2 * batch:
                return bd
    # blgnn.TrainStep: the iteration (sweep with gradients -> init_x -> pair batch -> loss -> backward -> SGD) captured in
    # a HIP graph and replayed per batch; --train-eager runs the identical iteration launch by launch
    ts = blgnn.TrainStep(model, data, opt, next_batch(), lower_cache=lower_cache, capture=captured)

    def step(_i):
        bd = next_batch()
        losses.append(ts.step(bd).clone())
        return len(bd.batch_gids)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scored = 0
    for i in range(steps):
        scored += step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    first, last = float(torch.stack(losses[:3]).mean()), float(torch.stack(losses[-3:]).mean())
    out = {"metric": "DDI edge-pairs/sec (bi-level TRAINING step: forward + backward + SGD)", "value": scored / el,
           "unit": "pairs/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_OF[transform],
           "data": "synthetic",
           "config": {"workload": ("TRAINING, DrugCombo-shape bi-level (lower GIN x5 + MetaLayer(GAT per edge type) x3, "
                                   "multi-label scorer, cross-entropy), d=128: sweep over all %d drugs with gradients + %d "
                                   "positive and as many sampled negative pairs per step" if combo else
                                   "TRAINING, DrugBank-shape DDI, lower GIN x5 + upper GAT x3, d=128: sweep over all %d "
                                   "drugs with gradients + %d positive and as many sampled negative pairs per step")
                                  % (cfg["drugs"], batch),
                      "atoms": int(lower_cache[0].merge_data['merge'].x.shape[0]), "optimizer": "SGD(momentum 0.9)",
                      "iteration": "captured in one HIP graph, replayed per batch (blgnn.TrainStep)" if captured
                                   else "eager (--train-eager)",
                      "loss_first3_mean": first, "loss_last3_mean": last}}
    if not args.no_cpu_baseline and not combo:
        # the oracle with torch autograd on the host cores: same model, same sizes, bounded number of steps
        cores = usable_cores()
        torch.set_num_threads(cores)
        specs = helpers.layer_specs(flags)
        of = dict(helpers.oracle_flags(flags, ds), train=True)
        pnames = {k for k, _ in model.named_parameters()}
        sd = {k: (v.clone().requires_grad_(k in pnames) if v.is_floating_point() else v)
              for k, v in helpers.state_dict_cpu(model).items()}
        lower_b = BatchData(all_molecule_pairs(ds.gs_map.keys()), ds, is_train=False, ignore_pairs=True)
        n, c0 = 0, time.perf_counter()
        while True:
            pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=batch, replace=False)]
            bd = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=rng)
            ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
            acts = oracle.run_layers(specs, sd, helpers.plain_batch(lower_b, ds, with_pairs=ids), of)
            loss = torch.nn.functional.binary_cross_entropy(acts[-1].view(-1), bd.true_labels().cpu().float())
            loss.backward()
            n += 1
            if time.perf_counter() - c0 > args.cpu_seconds or n >= 50:
                break
        ce = time.perf_counter() - c0
        out["cpu_baseline"] = {"value": n * 2 * batch / ce, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": "%d training steps (forward + torch autograd backward, no optimizer step) of "
                                         "the same model and sizes on the oracle, %.1f s" % (n, ce)}
    return out


def run_train_gmn(args, D, transform, steps, warmup):
    """Training step of the pair-batched GMN model (lower_level_gmn: GIN + GMNPropagator x3, src/config.py:329-350) at the
    BASELINE configs[2] / [4] shape, one GPU: per iteration a fresh pair batch (batch / 2 positives + as many sampled
    negatives, every molecule appended once per pair: src/batch.py:105-144), forward with train-mode BatchNorm, BCE,
    backward, SGD step -- what src/train.py:177-182 does.  Eager: the merged graph changes shape with every batch, so
    the iteration is not captured; every forward and backward step is a libblgnn kernel behind autograd nodes."""
    import numpy as np
    import torch
    import blgnn
    from blgnn.batch import BatchData
    cfg = CONFIGS[args.config]
    device = D.device
    d, pos_per_step = cfg["d"], cfg["batch"] // 2
    ds = blgnn.make_synthetic(cfg["drugs"], cfg["pairs"], shape=cfg["shape"], seed=SEED + 7, device=device)
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=d, device=device,
                             batch_size=cfg["batch"], feature_transform=transform)
    blgnn.set_flags(flags)
    torch.manual_seed(SEED)
    data = blgnn.TrainData(ds, device=device)
    model = blgnn.Model(data).to(device)
    model.train()
    params = [p for k, p in model.named_parameters() if k.startswith("layers.")]
    opt = torch.optim.SGD(params, lr=0.01, momentum=0.9)
    rng = np.random.default_rng(SEED)
    losses, atoms = [], []

    def step(_i):
        pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=pos_per_step, replace=False)]
        bd = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, unique_graphs=False, rng=rng)
        opt.zero_grad(set_to_none=True)
        loss = model(bd)
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
        atoms.append(int(bd.merge_data['merge'].x.shape[0]))
        return len(bd.batch_gids)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scored = 0
    for i in range(steps):
        scored += step(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    first, last = float(torch.stack(losses[:3]).mean()), float(torch.stack(losses[-3:]).mean())
    if not (first == first and last == last):
        raise SystemExit("bench.py --train: the loss is not finite")
    return {"metric": "DDI edge-pairs/sec (GMN TRAINING step: forward + backward + SGD)", "value": scored / el,
            "unit": "pairs/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * el / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_OF[transform], "data": "synthetic",
            "config": {"workload": "TRAINING, %s-shape lower-level GMN (GIN + GMNPropagator x3, pair batches), d=%d: %d positive "
                                   "and as many sampled negative pairs per step, a fresh batch (merge, CSR, segments) built on the "
                                   "host inside every timed step" % (cfg["shape"], d, pos_per_step),
                       "baseline_config_index": args.config, "atoms_per_step_mean": float(np.mean(atoms[-steps:])),
                       "optimizer": "SGD(momentum 0.9)", "iteration": "eager (the batch graph changes shape every step)",
                       "arithmetic": "y = x W^T and dx = g W on the forward path's fp32-equivalent 16-bit MFMA GEMMs (three fp16 / six bf16 "
                                     "products per fp32 product), weight gradients and small products on exact-f32 MFMA "
                                     "(blgnn/autograd.py)",
                       "loss_first3_mean": first, "loss_last3_mean": last}}


def main():
    args = parse()
    # --train without --config keeps timing the bi-level model (configs[1] shape); --train --config 2 / 4: the GMN model
    gmn_train = args.train and args.config is not None and CONFIGS[args.config]["kind"] == "gmn"
    if args.config is None:
        args.config = 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if os.environ.get("BLGNN_BENCH_RENDEZVOUS_ONLY"):
        return rendezvous_only(args)
    import torch                                            # noqa: F401  (first GPU-touching import of a rank)
    # host-side torch ops (the batch builder's index arithmetic) on the cores this process may use, shared among the ranks of
    # the node: by default torch sizes its intra-op pool by the VISIBLE cores (256 on a box that grants 16) and the
    # oversubscribed pool stalls for tens of ms at a time
    torch.set_num_threads(max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))))
    D = Dist(args)
    transform = "f32" if args.exact_f32 else args.transform
    if args.train:
        if D.world != 1:
            raise SystemExit("bench.py --train measures one GPU")
        print(json.dumps((run_train_gmn if gmn_train else run_train)(args, D, transform, args.steps, args.warmup)), flush=True)
        D.close()
        return 0
    cfg = CONFIGS[args.config]
    runner = run_bilevel if cfg["kind"] == "bilevel" else run_gmn
    res = runner(args.config, args, D, transform, args.steps, args.warmup, args.min_seconds)
    if D.rank == 0:
        out = {"metric": "DDI edge-pairs/sec (bi-level forward)",
               "value": res["value"], "unit": "pairs/s", "n_gpus": D.world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
               "higher_is_better": True, "scaling": cfg["scaling"], "vs_baseline": None,
               "dtype": DTYPE_OF[transform], "data": "synthetic",
               "timing": {"blocks_of_K_steps": res["timed_blocks"], "reported": "median block",
                          "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"]},
               "config": dict(res["config"], baseline_config_index=args.config),
               "roofline": res.get("roofline")}
        if res.get("interaction_gather"):
            out["interaction_gather"] = res["interaction_gather"]
        if res.get("value_incl_batch_build") is not None:
            out["value_incl_batch_build"] = res["value_incl_batch_build"]
        if not args.no_extra and D.world == 1:
            extra = {}

            def fresh_pools():
                # the extras run other configs in this process: hand the headline's cached allocator blocks back first, or
                # their eager (event-timed) passes pay hipMalloc / hipFree inside the timed launches (seen: 11 ms instead
                # of 4.2 ms per GIN launch of configs[3] behind configs[2]'s 100 GB of cached blocks)
                import gc
                gc.collect()
                torch.cuda.synchronize()
                torch.cuda.empty_cache()
            fresh_pools()
            if args.config != 1:
                # the round-1 headline configuration, for comparability across rounds
                r1 = run_bilevel(1, args, D, transform, 50, 10, 0.3, want_roofline=True)
                extra["configs_1"] = {"value": r1["value"], "unit": "pairs/s", "ms_per_step": r1["ms_per_step"],
                                      "workload": r1["config"]["workload"], "roofline": r1.get("roofline")}
            fresh_pools()
            if args.config != 3:
                # the bi-level path at scale on one GPU (BASELINE configs[3] at N = 1: 100 k drugs, 4 M atoms, d = 256)
                r3 = run_bilevel(3, args, D, transform, 10, 3, 0.2, want_roofline=True)
                extra["configs_3_n1"] = {"value": r3["value"], "unit": "pairs/s", "ms_per_step": r3["ms_per_step"],
                                         "workload": r3["config"]["workload"], "roofline": r3.get("roofline"),
                                         "interaction_gather": r3.get("interaction_gather")}
                # one rank's share of an N-GPU job at the same config, compute only (no collective runs; the interaction
                # level target-row sharded as it is with ranks): what the scaling projection in DESIGN section 6 rests on
                import copy
                emu_rows = {}
                for n_emu in (2, 4, 8):
                    fresh_pools()
                    a2 = copy.copy(args)
                    a2.emulate_world = n_emu
                    re_ = run_bilevel(3, a2, D, transform, 10, 3, 0.2, want_roofline=False)
                    emu_rows[str(n_emu)] = {"ms_per_step": re_["ms_per_step"],
                                            "interaction_level_ms": re_["config"]["interaction_level_ms"],
                                            "interaction_level": re_["config"].get("interaction_level")}
                extra["configs_3_emulated_ranks"] = {
                    "what": "bench.py --config 3 --emulate-world N: ONE rank's share of an N-GPU job on this GPU -- 1/N of the "
                            "molecules swept, its own target rows of the interaction level, 1/N of the pairs -- with every "
                            "all-gather replaced by a local copy of the rank's own block.  Compute-only projection, NOT a "
                            "multi-GPU measurement", "ranks": emu_rows}
            fresh_pools()
            r5 = run_bilevel(5, args, D, transform, 50, 10, 0.3, want_roofline=False)
            extra["drugcombo_interaction_level"] = {
                "workload": r5["config"]["workload"], "ms_per_step": r5["ms_per_step"], "value": r5["value"], "unit": "pairs/s",
                "interaction_level_ms": r5["config"]["interaction_level_ms"],
                "what": "interaction_level_ms = LoadInteraction + 3 x MetaLayerWrapper(NodeModelAggrByEdge('gat'): one GATConv "
                        "per edge type, summed in the aggregation epilogue) + multi-label pair scorer on %d interaction nodes, "
                        "captured graph replayed back to back" % r5["config"]["drugs"]}
            fresh_pools()
            extra.update(micro_roofline(D.device))
            out["extra"] = extra
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = res["_cpu"](args.cpu_seconds)
        print(json.dumps(out), flush=True)
    D.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
