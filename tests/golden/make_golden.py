#!/usr/bin/env python3
"""Generates tests/golden/drugs_small.npz.

Inputs: the 20 real DrugBank molecules the reference ships under
data/DrugBank/ddi_data/drugs_small/*.gexf (517 atoms), parsed with the stdlib XML reader
in blgnn/io_gexf.py and one-hot encoded by the reference's rule
(utils/data/representation_node_feat.py:10-33,86-97).  These are DATA files of the
reference, not source.

Expected outputs: produced by THIS repo's CPU oracle in float64 (the reference cannot be
imported here: torch_geometric / torch_scatter / torch_sparse / klepto are absent and
un-vendored -- SURVEY section 8c -- so parity is "unpinned": the fixture pins the oracle
and the HIP path against regressions, and carries real-molecule inputs to the GPU box,
where /root/reference does not exist).

Weights come from a fixed torch seed but are STORED in the fixture, so the test does not
depend on torch's RNG stream.  Run from the repo root in the build container:
    python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "bilevel-graph-neural-network_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import blgnn                                            # noqa: E402
import oracle                                           # noqa: E402
from blgnn.io_gexf import load_gexf_dir                 # noqa: E402
from blgnn.graph import MoleculeStore                   # noqa: E402
import helpers                                          # noqa: E402

SRC = "/root/reference/data/DrugBank/ddi_data/drugs_small"
D = 32


def main():
    gids, xs, eis = load_gexf_dir(SRC)
    store = MoleculeStore.from_lists(xs, eis, gids, "cpu")
    rng = np.random.default_rng(7)
    n = len(gids)
    # a small interaction graph over the 20 drugs: 45 distinct pairs
    all_pairs = np.asarray([(gids[i], gids[j]) for i in range(n) for j in range(i + 1, n)])
    pairs = all_pairs[rng.choice(len(all_pairs), size=45, replace=False)]
    ds = blgnn.BiGNNDataset("drugs_small", store, pairs, np.ones(len(pairs), dtype=np.int64))
    out = {"gids": np.asarray(gids), "x": store.x.numpy(), "edge_local": store.edge_local.numpy(),
           "node_ptr": store.node_ptr, "edge_ptr": store.edge_ptr, "pairs": pairs}

    def run(tag, flags, unique=True, batch_pairs=None):
        model, data = helpers.build(flags, ds, seed=11)
        specs = helpers.layer_specs(flags)
        of = helpers.oracle_flags(flags, ds)
        if flags.lower_level_layers and flags.higher_level_layers:
            from blgnn.forward import all_molecule_pairs
            k = flags.last_lower_lyr_num
            lb = blgnn.BatchData(all_molecule_pairs(gids), ds, is_train=False, ignore_pairs=True)
            pl32 = helpers.plain_batch(lb, ds)
            sd = helpers.calibrate_bn(model, specs[:k + 1], pl32, of)
            oracle.run_layers(specs[:k + 1], sd, pl32, of)
            hb = blgnn.BatchData(batch_pairs, ds, is_train=False)
            ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(hb.batch_gids))
            ph32 = helpers.plain_batch(hb, ds, init_x=pl32["init_x"], with_pairs=ids)
            sd = helpers.calibrate_bn(model, specs[k + 1:], ph32, of, first=k + 1)
            sd64 = {a: (b.double() if b.is_floating_point() else b) for a, b in sd.items()}
            pl = helpers.plain_batch(lb, ds, torch.float64)
            lower = oracle.run_layers(specs[:k + 1], sd64, pl, of)
            ph = helpers.plain_batch(hb, ds, torch.float64, init_x=pl["init_x"], with_pairs=ids)
            upper = oracle.run_layers(specs[k + 1:], sd64, ph, of, first=k + 1)
            acts = lower[1:] + upper[1:]
        else:
            bd = blgnn.BatchData(batch_pairs, ds, is_train=False, unique_graphs=unique)
            ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids)) \
                if unique else torch.arange(2 * len(batch_pairs)).reshape(-1, 2)
            pl32 = helpers.plain_batch(bd, ds, with_pairs=ids)
            sd = helpers.calibrate_bn(model, specs, pl32, of)
            sd64 = {a: (b.double() if b.is_floating_point() else b) for a, b in sd.items()}
            acts = oracle.run_layers(specs, sd64, helpers.plain_batch(bd, ds, torch.float64, with_pairs=ids), of)[1:]
        out[tag + "/specs"] = np.asarray(specs)
        for name, v in sd.items():
            if name.startswith("layers."):
                out[tag + "/sd/" + name] = v.numpy()
        for i, a in enumerate(acts):
            out[tag + "/act/%02d" % i] = a.numpy()
        out[tag + "/batch_pairs"] = np.asarray(batch_pairs)

    for group in ("src", "dst"):
        run("bignn_gin_gat_" + group,
            blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=D, D_higher=D, device="cpu",
                             gat_softmax_group=group), batch_pairs=pairs[:16])
    run("llgnn_gcn", blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gcn", D_lower=D, device="cpu"),
        batch_pairs=pairs[:12])
    run("mhcaddi_gmn", blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=D, device="cpu"),
        unique=False, batch_pairs=pairs[:10])
    path = os.path.join(HERE, "drugs_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "arrays")


if __name__ == "__main__":
    main()
