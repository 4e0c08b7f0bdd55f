"""GPU parity AT THE BENCH SIZES of BASELINE.json configs[2], [3] and [4] (the sizes `bench.py` times), through a property
of the path: in eval mode pairs (GMN models) and molecules (the lower sweep) are independent connected components --
BatchNorm is a per-channel affine, the cross-graph attention couples only the two graphs of a pair -- so the rows the
full-size GPU forward produced for a random subset of pairs / molecules must equal the CPU oracle run on THAT SUBSET
ALONE, which takes seconds.  Every layer's rows of the subset are compared, not just the final scores.

Each comparison also evaluates the oracle in fp32 on the CPU (the arithmetic of the reference's own CPU path) against the
fp64 oracle: ``cpu32_*`` in gpurun_out/parity_errors.json beside the GPU's error, and the GPU is held to
max(2 x the fp32 CPU error, 4e-7 of the tensor's magnitude) -- i.e. "as good as running the reference's fp32 arithmetic
in another summation order"."""

import copy

import numpy as np
import pytest
import torch

import blgnn
import oracle
from blgnn.forward import get_initial_embd, all_molecule_pairs
from helpers import (TOL, compare3, build, layer_specs, plain_batch, oracle_flags, calibrate_bn)

pytestmark = pytest.mark.gpu

FLOOR = 4e-7       # a few fp32 ulps of the tensor's largest entry: below this, ratios of rounding noise mean nothing


def _check3(got, ref64, cpu32, what, ratio=2.0):
    g, c = compare3(got, ref64, cpu32)
    assert g < TOL, (what, g)
    assert g <= max(ratio * c, FLOOR), (what, "GPU %.3e vs fp32 CPU oracle %.3e (scaled)" % (g, c))


def _gmn_full_size(device, d, drugs, npairs, B, seed, dataset, edge_types, shape, subset=256, ratio=2.0):
    ds = blgnn.make_synthetic(drugs, npairs, shape=shape, seed=seed, device=device, num_edge_types=edge_types)
    flags = blgnn.make_flags("lower_level_gmn", dataset=dataset, lower_level_gnn_type="gin", D_lower=d,
                             device=device, batch_size=B)
    model, data = build(flags, ds, seed=seed + 1, num_hyper_edge_feat=3 if edge_types > 1 else 0)
    specs = layer_specs(flags)
    rng = np.random.default_rng(seed + 2)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=B, replace=False)]
    sel = np.sort(rng.choice(B, size=subset, replace=False))
    bd_sub = blgnn.BatchData(pairs[sel], ds, is_train=False, unique_graphs=False)
    ids = torch.arange(2 * subset).reshape(-1, 2)
    of = oracle_flags(flags, ds)
    # eval-BN statistics of a trained model: calibrated on the subset (the full batch has the same distribution)
    sd32 = calibrate_bn(model, specs, plain_batch(bd_sub, ds, torch.float32, with_pairs=ids), of)
    sd64 = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts64 = oracle.run_layers(specs, sd64, plain_batch(bd_sub, ds, torch.float64, with_pairs=ids), of)
    acts32 = oracle.run_layers(specs, sd32, plain_batch(bd_sub, ds, torch.float32, with_pairs=ids), of)
    # the full-size forward, as bench.py runs it
    bd = blgnn.BatchData(pairs, ds, is_train=False, unique_graphs=False)
    model.acts = [bd.merge_data['merge'].x]
    with torch.no_grad():
        for layer in model.layers[:-1]:
            model.acts.append(layer(model.acts[-1], bd, model))
    torch.cuda.synchronize()
    Nn = bd.merge_data['merge'].x.shape[0]
    ind = bd.merge_data['ind_list']
    graphs = np.stack([2 * sel, 2 * sel + 1], axis=1).reshape(-1)             # pair i = graphs 2 i, 2 i + 1
    node_rows = torch.as_tensor(np.concatenate([np.arange(ind[g][0], ind[g][1]) for g in graphs]), device=device)
    graph_rows = torch.as_tensor(graphs, device=device)
    pair_rows = torch.as_tensor(sel, device=device)
    b_sub = bd_sub.merge_data['merge'].batch.cpu()
    assert len(model.acts) == len(acts64)
    for k, (got, r64, r32) in enumerate(zip(model.acts[1:], acts64[1:], acts32[1:])):
        kind = getattr(got, "_blgnn_pooled", None)
        if kind is not None:            # the last propagator's rows arrive pooled per graph (layers_gmn.py ``_readout_of``)
            G = 2 * subset
            pool = lambda r: (torch.zeros((G, r.shape[1]), dtype=r.dtype).index_add_(0, b_sub, r) /
                              (torch.bincount(b_sub, minlength=G).clamp(min=1).view(-1, 1).to(r.dtype) if kind == "mean" else 1))
            r64, r32 = pool(r64), pool(r32)
        rows = node_rows if got.shape[0] == Nn else (graph_rows if got.shape[0] == 2 * B else pair_rows)
        assert got.shape[0] in (Nn, 2 * B, B)
        _check3(got.index_select(0, rows), r64, r32, "layer %d (%s)" % (k, specs[k].split(":")[0]), ratio)
    return Nn


def test_config2_at_bench_size_subset_oracle(device):
    """configs[2] as benchmarked: DrugCombo shape, 8 192 pairs (16 384 molecule instances, ~480 k atoms), GIN + GMN x3,
    d = 256, shared prefix + pooled tail on; 256 random pairs of the batch against the oracle on those pairs alone"""
    Nn = _gmn_full_size(device, 256, 3376, 50263, 8192, 20200617, "drugcombo", 2, "drugcombo")
    assert Nn > 400000


def test_config4_at_bench_size_subset_oracle(device):
    """configs[4] as benchmarked on one GPU: d = 512, 8 192 pairs of ~40-atom molecules (~657 k atoms, 5.4 GB GEMM operands:
    the sizes where 32-bit element offsets would wrap); 128 random pairs against the oracle.  Bound 3 x the fp32 CPU
    oracle instead of 2 x: f_node's K = 2 048 products run as ONE fp32 accumulation chain per output inside the matrix
    cores, the CPU's blocked GEMM sums 8-16 partial chains (measured: GPU 1.3e-6 vs CPU 6.1e-7 of the largest entry)"""
    Nn = _gmn_full_size(device, 512, 100000, 1000000, 8192, 20200619, "drugbank", 1, "atoms40", subset=128, ratio=3.0)
    assert Nn > 600000


def test_config3_at_bench_size_init_x_of_random_molecules(device):
    """configs[3] as benchmarked at N = 1: the lower sweep over 100 k molecules (4 M atoms, GIN x5 at d = 256, multi-scale
    readout fused into the layers) -> init_x [100 k, 1 280]; the rows of 200 random molecules against the oracle run on
    those molecules alone"""
    ds = blgnn.make_synthetic(100000, 1000000, shape="atoms40", seed=20200618, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=256, D_higher=256, device=device)
    model, data = build(flags, ds, seed=61)
    specs = layer_specs(flags)
    k = flags.last_lower_lyr_num
    rng = np.random.default_rng(62)
    gids = np.asarray(ds.molecules.gids)
    sel = np.sort(rng.choice(len(gids), size=200, replace=False))
    sub = blgnn.BatchData(all_molecule_pairs(gids[sel]), ds, is_train=False, ignore_pairs=True)
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs[:k + 1], plain_batch(sub, ds, torch.float32), of)
    sd64 = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    p64, p32 = plain_batch(sub, ds, torch.float64), plain_batch(sub, ds, torch.float32)
    oracle.run_layers(specs[:k + 1], sd64, p64, of)
    oracle.run_layers(specs[:k + 1], sd32, p32, of)
    rows = torch.as_tensor([ds.gs_map[g] for g in gids[sel]])
    with torch.no_grad():
        get_initial_embd(data, model)
    torch.cuda.synchronize()
    init_x = ds.interaction_combo_nxgraph.init_x
    assert init_x.shape == (100000, 1280)
    _check3(init_x.index_select(0, rows.to(device)), p64["init_x"][rows], p32["init_x"][rows], "init_x rows")
