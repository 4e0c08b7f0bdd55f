#!/usr/bin/env python3
"""Headline benchmark: DDI edge-pairs/sec of the bi-level forward on N MI355X.

Workload (BASELINE.json configs[1]): DrugBank-shaped synthetic data -- 1 326 drugs per GPU
(log-normal molecule sizes, ~28 atoms, F_in = 52), interaction graph with 29 240 train pairs
per 1 326 drugs, lower GIN x5 (d=128) + multi-scale avg-pool readout + upper GAT x3 +
LinkPred; 4 096 edge pairs scored per GPU per step.

One step = one complete bi-level forward (src/train.py:75-108 of the reference):
  lower sweep over this rank's molecules -> readout -> [all-gather of pooled drug
  embeddings over RCCL when N > 1] -> interaction-level GAT x3 -> pair scores for this
  rank's 4 096 pairs.  Inputs (features, edge lists, pair ids) are resident in HBM before
  the timed region; the molecule set and the interaction graph are static, so their CSR
  layouts are built once (the reference rebuilds both per forward).

Weak scaling: per-GPU work is fixed; value = N * 4096 * steps / max-over-ranks time.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the
dominant kernel (fp32-MFMA feature transform of the lower level, timed live with HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample).
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "bilevel-graph-neural-network_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

DRUGS_PER_GPU = 1326
PAIRS_PER_DRUGSET = 29240      # 70 % train split of DrugBank's 41 772 pairs
BATCH_PAIRS = 4096
D = 128
SEED = 20200614 + 2
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA (same table)
PEAK_HBM_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the d=256 SpMM/readout micro-roofline")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="profiling aid: build rank 0's share of an N-GPU run on one GPU (no collective)")
    ap.add_argument("--fuse-lower", action="store_true",
                    help="run the lower level as ONE launch (blgnn_gin_stack) instead of one per GIN layer")
    ap.add_argument("--transform", choices=("f16x3", "bf16x6", "f32"), default="f16x3",
                    help="how the fused GIN layer runs its dense transforms: fp16 two-piece split (default), "
                         "bf16 three-piece split, or exact-f32 MFMA -- all fp32-equivalent")
    ap.add_argument("--exact-f32", action="store_true", help="same as --transform f32")
    ap.add_argument("--lower-cus", type=int, default=-1,
                    help="compute units per XCD (of 32) reserved for the molecular sweep's stream; the "
                         "interaction-level stream gets the rest.  0 = ordinary streams; -1 = the measured optimum "
                         "for this workload (20 at 1 GPU, 16 at 2, none from 4 GPUs on, where the replicated "
                         "interaction level wants the whole chip)")
    ap.add_argument("--readout-with-upper", action="store_true",
                    help="move the node->graph readout from the tail of the molecular sweep's graph to the head of "
                         "the interaction-level graph (two-stream balance knob; slower at the DrugBank shape)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="replay the lower and upper graphs of a step back to back on one stream")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying captured HIP graphs")
    return ap.parse_args()


class KernelTimer(object):
    """HIP-event timing of one op's launches on the stream they are launched on
    (torch's current stream == the stream handed to libblgnn)."""

    def __init__(self):
        self.pairs, self.work, self.enabled = [], [], False

    def wrap(self, fn, work_of, select):
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
            return out
        return wrapped

    def result(self):
        ms = [a.elapsed_time(b) for a, b in self.pairs]
        return ms, self.work


def build_workload(rank, world, device, transform="f16x3"):
    import blgnn
    from blgnn.dist import ShardedLower
    n_drugs = DRUGS_PER_GPU * world
    ds = blgnn.make_synthetic(n_drugs, PAIRS_PER_DRUGSET * world, shape="drugbank", seed=SEED, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=D, D_higher=D, device=device,
                             batch_size=BATCH_PAIRS, feature_transform=transform)
    blgnn.set_flags(flags)
    torch.manual_seed(SEED)
    data = blgnn.TrainData(ds, device=device)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    model = blgnn.Model(data)
    g = torch.Generator().manual_seed(SEED + 1)
    with torch.no_grad():                      # eval BN, running stats perturbed by U(-0.1, 0.1)
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(0.2 * (torch.rand(mod.running_mean.shape, generator=g) - 0.5))
                mod.running_var.copy_(1.0 + 0.2 * (torch.rand(mod.running_var.shape, generator=g) - 0.5))
    model.to(device).eval()
    sharded = ShardedLower(ds.molecules.gids, ds.molecules.sizes, rank, world)
    return ds, flags, data, model, sharded


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; the hot path has no CPU fallback")
    # BLGNN_BENCH_BACKEND=gloo rehearses the N > 1 code path on a box with fewer GPUs than ranks
    backend = os.environ.get("BLGNN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import blgnn
    from blgnn import ops
    from blgnn.batch import BatchData
    from blgnn.forward import all_molecule_pairs

    transform = "f32" if args.exact_f32 else args.transform
    if args.emulate_world > 1 and world == 1:
        ds, flags, data, model, sharded = build_workload(0, args.emulate_world, device, transform)
    else:
        ds, flags, data, model, sharded = build_workload(rank, world, device, transform)
    k = flags.last_lower_lyr_num
    lower_layers = list(model.layers[:k + 1])
    upper_layers = list(model.layers[k + 1:-1])          # LoadInteraction, GAT x3, LinkPred (no Loss)

    # -- static inputs, resident in HBM before timing -------------------------------------
    lower_batch = BatchData(all_molecule_pairs(sharded.local_gids), ds, is_train=False, ignore_pairs=True)
    lower_batch.merge_data['merge'].csr()
    graph = ds.interaction_combo_nxgraph
    d_pool = D * 5
    graph.init_x = torch.zeros((ds.num_graphs(), d_pool), dtype=torch.float32, device=device)
    ds.init = True
    rng = np.random.default_rng(SEED + 17 + rank)
    n_batches = 8
    upper_batches = []
    for _ in range(n_batches):
        idx = rng.choice(len(ds.pair_gids), size=BATCH_PAIRS, replace=False)
        b = BatchData(ds.pair_gids[idx], ds, is_train=False)
        upper_batches.append(b)
    Nn = int(lower_batch.merge_data['merge'].x.shape[0])
    nnz = int(lower_batch.merge_data['merge'].edge_index.shape[1])

    # dominant kernel: the fused GIN layer (CSR gather -> LDS -> 2 matrix-core GEMMs -> epilogue),
    # bf16x6 variant by default (FLAGS.feature_transform), exact-f32 MFMA variant otherwise
    timer = KernelTimer()
    real_gin, real_gin6, real_gin3 = ops.gin_layer, ops.gin_layer_x6, ops.gin_layer_h3
    ops.gin_layer = timer.wrap(
        real_gin,
        lambda csr, x, w1, b1, w2, *a, **kw: (2.0 * x.shape[0] * (w1.shape[0] * w1.shape[1] + w2.shape[0] * w2.shape[1]),
                                              4.0 * x.shape[0] * (x.shape[1] + w1.shape[1]) + 4.0 * csr.nnz + 4.0 * (csr.n + 1)),
        lambda *a, **kw: True)
    ops.gin_layer_x6 = timer.wrap(
        real_gin6,
        lambda csr, x, w1p, b1, w2p, b2, N, *a, **kw: (2.0 * x.shape[0] * (x.shape[1] * N + N * N),
                                                       4.0 * x.shape[0] * (x.shape[1] + N) + 4.0 * csr.nnz + 4.0 * (csr.n + 1)),
        lambda *a, **kw: True)
    ops.gin_layer_h3 = timer.wrap(
        real_gin3,
        lambda csr, x, w1p, b1, w2p, b2, N, *a, **kw: (2.0 * x.shape[0] * (x.shape[1] * N + N * N),
                                                       4.0 * x.shape[0] * (x.shape[1] + N) + 4.0 * csr.nnz + 4.0 * (csr.n + 1)),
        lambda *a, **kw: True)
    mode = getattr(flags, "feature_transform", "f16x3")

    # The forward is ~35 short launches: captured into HIP graphs and replayed, and consecutive
    # forwards are software-pipelined over two HIP streams (blgnn/engine.py): the molecular
    # sweep [+ RCCL all-gather, kept outside capture] of step k+1 overlaps the interaction-level
    # pass + pair scorer of step k.  --no-graph / --no-overlap switch either off.
    from blgnn.engine import BiLevelEngine
    eff_world = args.emulate_world if (args.emulate_world > 1 and world == 1) else world
    lower_cus = {1: 20, 2: 16}.get(eff_world, 0) if args.lower_cus < 0 else args.lower_cus
    engine = BiLevelEngine(model, lower_batch, upper_batches, graph, sharded=sharded, world_size=world,
                           use_graph=not args.no_graph, overlap=not args.no_overlap,
                           fuse_lower=args.fuse_lower, readout_with_upper=args.readout_with_upper,
                           lower_cus_per_xcd=lower_cus)
    use_graph = engine.use_graph
    engine.capture()
    step = engine.eager_step
    run = engine.step
    for i in range(args.warmup):
        run(i)
    engine.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer.enabled = not use_graph
    for i in range(args.steps):
        preds = run(i)
    engine.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if use_graph:
        # HIP events cannot sit inside a replayed graph: the dominant kernel is timed over the
        # same K steps launched eagerly right after the timed region (same inputs, same stream)
        # A long matmul is queued ahead of every step so that the step's launches wait in the
        # queue and the events bracket back-to-back execution, not host launch latency.
        timer.enabled = True
        blk = torch.empty(4096, 8192, device=device)
        blk_w = torch.empty(8192, 8192, device=device)
        for i in range(args.steps):
            torch.mm(blk, blk_w)
            step(i)
        torch.cuda.synchronize()
        timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert preds.shape == (BATCH_PAIRS, 1) and bool(torch.isfinite(preds).all())

    ms, work = timer.result()
    ops.gin_layer, ops.gin_layer_x6, ops.gin_layer_h3 = real_gin, real_gin6, real_gin3
    out = None
    if rank == 0:
        value = world * BATCH_PAIRS * args.steps / elapsed
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (profiles/summarize_pmc.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction)
        traffic = None
        try:
            name = {"f16x3": "r01_gin_layer_h3_pmc_summary.json", "bf16x6": "r01_gin_layer_x6_pmc_summary.json",
                    "f32": "r01_gin_layer_pmc_summary.json"}[mode]
            traffic = json.load(open(os.path.join(ROOT, "profiles", name)))["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        tflops = sum(w[0] for w in work) / (sum(ms) * 1e-3) / 1e12 if ms else 0.0
        alg_gbps = sum(w[1] for w in work) / (sum(ms) * 1e-3) / 1e9 if ms else 0.0
        # split transforms: every algorithmic fp32 product is three fp16 (f16x3) or six bf16 (bf16x6) MFMA
        # products -> the executed matrix-core rate is 3x / 6x the algorithmic one and is priced against the
        # dense fp16 / bf16 peak (the same 2.5 PFLOP/s)
        mfma_mult, peak = {"f16x3": (3.0, PEAK_BF16_MFMA_TFLOPS), "bf16x6": (6.0, PEAK_BF16_MFMA_TFLOPS),
                           "f32": (1.0, PEAK_F32_MFMA_TFLOPS)}[mode]
        kname = {"f16x3": "k_gin_layer_h3 (fused GIN layer: CSR gather in LDS + 2 GEMMs as 3 fp16 MFMA products per "
                          "fp32 product + act/BN)",
                 "bf16x6": "k_gin_layer_x6 (fused GIN layer: CSR gather in LDS + 2 GEMMs as 6 bf16 MFMA products per "
                           "fp32 product + act/BN)",
                 "f32": "k_gin_layer (fused GIN layer: CSR gather in LDS + 2 exact-f32 MFMA GEMMs + act/BN)"}[mode]
        # The binding roofline of the launch is whichever floor is higher: algorithmic bytes at the HBM
        # peak, or EXECUTED matrix-core flops (3x / 6x / 1x the algorithmic fp32 flops) at the dense peak of
        # the MFMA dtype.  f16x3 halves the matrix-core work of bf16x6, which leaves HBM as the floor at d=128.
        exec_tflops = mfma_mult * tflops
        mfma_frac, hbm_frac = exec_tflops / peak, alg_gbps / PEAK_HBM_GBPS
        by_hbm = hbm_frac >= mfma_frac
        roofline = {
            "kernel": kname,
            "bound": "hbm" if by_hbm else "mfma",
            "achieved": alg_gbps if by_hbm else exec_tflops,
            "peak": PEAK_HBM_GBPS if by_hbm else peak,
            "unit": "GB/s" if by_hbm else "TFLOP/s",
            "frac": hbm_frac if by_hbm else mfma_frac, "traffic": traffic,
            "alg_bytes_per_launch": sum(w[1] for w in work) / max(len(work), 1),
            "alg_flops_per_launch": sum(w[0] for w in work) / max(len(work), 1),
            "launches": len(ms), "avg_us": 1e3 * sum(ms) / max(len(ms), 1),
            "alg_hbm_GBps": alg_gbps, "alg_hbm_frac": hbm_frac,
            "executed_mfma_tflops": exec_tflops, "executed_mfma_frac": mfma_frac, "mfma_peak_tflops": peak,
            "algorithmic_f32_tflops": tflops, "frac_of_f32_mfma_peak": tflops / PEAK_F32_MFMA_TFLOPS}
        out = {
            "metric": "DDI edge-pairs/sec (bi-level forward)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"f16x3": "f32 (dense transforms as f16x3 split MFMA, fp32-equivalent)",
                      "bf16x6": "f32 (dense transforms as bf16x6 split MFMA, fp32-equivalent)", "f32": "f32"}[mode],
            "data": "synthetic",
            "config": {"workload": "DrugBank-shape DDI, lower GIN x5 + upper GAT x3, d=128, "
                                   "batch=4096 edge pairs per GPU (BASELINE configs[1])",
                       "drugs_per_gpu": DRUGS_PER_GPU, "atoms_per_gpu": Nn, "mol_edges_per_gpu": nnz,
                       "interaction_pairs": int(len(ds.pair_gids)), "d": D,
                       "pairs_per_step_per_gpu": BATCH_PAIRS, "hip_graph": bool(use_graph),
                       "two_stream_overlap": bool(engine.overlap),
                       "readout_with_upper": bool(engine.readout_with_upper),
                       "lower_cus_per_xcd": engine.lower_cus_per_xcd,
                       "fused_lower_stack": engine.fused is not None,
                       "parallelism": "molecule/pair shards x%d + all-gather(pooled)" % world},
            "roofline": roofline,
        }
        if not args.no_extra:
            out["extra"] = micro_roofline(device)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ds, flags, model, lower_batch, upper_batches[0], args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def micro_roofline(device):
    """North-star kernel evidence: molecular SpMM (GIN aggregation) and segment-mean readout at
    d=256 on a 12 500-molecule x ~40-atom shard (BASELINE configs[3] per-GPU share), HIP-event
    timed; algorithmic bytes per SURVEY section 8(d)."""
    import blgnn
    from blgnn import ops
    ds = blgnn.make_synthetic(12500, 10, shape="atoms40", seed=SEED + 100, device=device)
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
                        "frac": b / t / 1e9 / PEAK_HBM_GBPS, "alg_bytes": b, "us": t * 1e6}
    t = timeit(lambda: ops.segment_reduce(seg, x, "mean", out=pooled))
    b = 4.0 * d * n + 4.0 * d * G + 4.0 * (G + 1)
    res["readout_d256"] = {"bound": "hbm", "achieved": b / t / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                           "frac": b / t / 1e9 / PEAK_HBM_GBPS, "alg_bytes": b, "us": t * 1e6}
    w = torch.randn(d, d, device=device) / 16
    h = torch.empty_like(x)
    t = timeit(lambda: ops.linear(x, w, out=h, transform="f32"), reps=10)
    f = 2.0 * n * d * d
    res["linear_d256"] = {"bound": "mfma", "achieved": f / t / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                          "unit": "TFLOP/s", "frac": f / t / 1e12 / PEAK_F32_MFMA_TFLOPS, "us": t * 1e6}
    # same GEMM as six bf16 MFMA products per fp32 product: executed rate against the bf16 peak, and
    # the launch against HBM (X read once, Y written once)
    t = timeit(lambda: ops.linear(x, w, out=h, transform="bf16x6"), reps=10)
    by = 8.0 * n * d
    res["linear_d256_bf16x6"] = {"bound": "mfma", "achieved": 6.0 * f / t / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS,
                                 "unit": "TFLOP/s", "frac": 6.0 * f / t / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                                 "algorithmic_f32_tflops": f / t / 1e12, "alg_hbm_GBps": by / t / 1e9,
                                 "alg_hbm_frac": by / t / 1e9 / PEAK_HBM_GBPS, "us": t * 1e6}
    return res


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


def cpu_baseline(ds, flags, model, lower_batch, upper_batch, seconds):
    """The CPU oracle (pure PyTorch restatement of the reference path) on the host cores, same
    workload, same weights; bounded sample of whole steps."""
    import oracle
    import helpers
    cores = usable_cores()
    torch.set_num_threads(cores)
    specs = helpers.layer_specs(flags)
    k = flags.last_lower_lyr_num
    sd = helpers.state_dict_cpu(model)
    of = helpers.oracle_flags(flags, ds)
    pl = helpers.plain_batch(lower_batch, ds)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(upper_batch.batch_gids))
    ph0 = helpers.plain_batch(upper_batch, ds, with_pairs=ids)     # static graph layouts built once
    n = 0
    t0 = time.perf_counter()
    with torch.no_grad():
        while True:
            p = dict(pl)
            p["init_x"] = None
            oracle.run_layers(specs[:k + 1], sd, p, of)
            ph = dict(ph0)
            ph["init_x"] = p["init_x"]
            oracle.run_layers(specs[k + 1:], sd, ph, of, first=k + 1)
            n += 1
            if time.perf_counter() - t0 > seconds or n >= 200:
                break
    dt = time.perf_counter() - t0
    return {"value": n * BATCH_PAIRS / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d whole steps of the same workload (oracle, torch CPU fp32, %d threads), %.1f s"
                      % (n, cores, dt)}


if __name__ == "__main__":
    main()
