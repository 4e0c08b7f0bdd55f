"""GPU parity at the REAL DrugBank shape: the 1 309 molecules (36 816 atoms) and 41 072 interaction pairs of
the reference's data directory, committed as a data-only fixture in compact code form
(tests/golden/drugbank_real.npz, built by tests/golden/make_drugbank.py), expanded to one-hot rows on the
device, run through BASELINE configs[0] (d=64, 32 pairs) and configs[1] (d=128, 4 096 pairs) against the
fp64 oracle."""

import os

import numpy as np
import pytest
import torch

import blgnn
from blgnn import io_raw
from blgnn.forward import bilevel_forward
from helpers import TOL, max_err, abs_err, build, layer_specs
from test_gpu_model import _oracle_bilevel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drugbank_real.npz")


def _dataset(device):
    z = np.load(GOLD)
    return io_raw.dataset_from_codes(z["gids"].tolist(), z["node_ptr"], z["codes"], z["cards"], z["edge_ptr"],
                                     z["edge_local"], z["pairs"], device=device)


@pytest.mark.parametrize("d,B,transform", [(64, 32, "f16x3"), (128, 4096, "f16x3"), (128, 4096, "bf16x6")])
def test_real_drugbank_bilevel_forward(device, d, B, transform):
    ds = _dataset(device)
    assert ds.num_graphs() == 1309 and ds.molecules.num_node_feat == 52
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=d, D_higher=d, device=device,
                             feature_transform=transform)
    model, data = build(flags, ds, seed=61)
    specs = layer_specs(flags)
    rng = np.random.default_rng(5)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=B, replace=False)]
    init_ref, acts_ref = _oracle_bilevel(model, flags, ds, specs, pairs)
    preds, bd = bilevel_forward(model, data, pairs)
    assert max_err(ds.interaction_combo_nxgraph.init_x, init_ref) < TOL
    for got, ref in zip(model.acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < TOL
    assert abs_err(preds, acts_ref[-1]) < 1e-5                     # sigmoid scores: the north-star bound literally
