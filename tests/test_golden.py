"""Golden fixture (tests/golden/drugs_small.npz): the reference's 20 real drugs_small
molecules as inputs, stored weights, expected activations from the float64 oracle.
CPU: the oracle reproduces the fixture.  GPU: the HIP path matches it within 1e-5."""

import os

import numpy as np
import pytest
import torch

import blgnn
import oracle
from blgnn.graph import MoleculeStore
from blgnn.forward import all_molecule_pairs, bilevel_forward
import helpers
from helpers import TOL, max_err

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "drugs_small.npz")
TAGS = {
    "bignn_gin_gat_src": dict(model="lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                              higher_level_gnn_type="gat", D_lower=32, D_higher=32, gat_softmax_group="src"),
    "bignn_gin_gat_dst": dict(model="lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                              higher_level_gnn_type="gat", D_lower=32, D_higher=32, gat_softmax_group="dst"),
    "llgnn_gcn": dict(model="lower_level_gnn", lower_level_gnn_type="gcn", D_lower=32),
    "mhcaddi_gmn": dict(model="lower_level_gmn", lower_level_gnn_type="gin", D_lower=32),
}


@pytest.fixture(scope="module")
def fx():
    return np.load(PATH, allow_pickle=False)


def _dataset(fx, device):
    store = MoleculeStore(fx["x"], fx["edge_local"], fx["node_ptr"], fx["edge_ptr"], fx["gids"].tolist(), device)
    return blgnn.BiGNNDataset("drugs_small", store, fx["pairs"], np.ones(len(fx["pairs"]), dtype=np.int64))


def _setup(fx, tag, device):
    ds = _dataset(fx, device)
    flags = blgnn.make_flags(device=device, **TAGS[tag])
    model, data = helpers.build(flags, ds, seed=0)
    sd = {k[len(tag) + 4:]: torch.as_tensor(fx[k]) for k in fx.files if k.startswith(tag + "/sd/")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(not m.startswith("layers.") for m in missing)
    specs = [str(s) for s in fx[tag + "/specs"]]
    assert specs == helpers.layer_specs(flags)          # the spec builder reproduces the fixture's strings
    acts = [torch.as_tensor(fx[k]) for k in sorted(f for f in fx.files if f.startswith(tag + "/act/"))]
    return ds, flags, model, data, specs, acts, fx[tag + "/batch_pairs"]


def test_fixture_inputs_are_the_drugs_small_molecules(fx):
    assert fx["x"].shape == (517, 18) and len(fx["gids"]) == 20            # 20 molecules, 517 atoms
    assert (fx["x"].sum(1) == 6).all()                                     # six one-hot attributes per atom
    ds = _dataset(fx, "cpu")
    for gid in ds.molecules.gids:
        assert ds.molecules.single(gid).is_undirected()


@pytest.mark.parametrize("tag", sorted(TAGS))
def test_oracle_reproduces_fixture(fx, tag):
    ds, flags, model, data, specs, acts, bp = _setup(fx, tag, "cpu")
    of = helpers.oracle_flags(flags, ds)
    for dtype, tol in ((torch.float64, 1e-10), (torch.float32, TOL)):
        sd = helpers.state_dict_cpu(model, dtype)
        if flags.higher_level_layers:
            k = flags.last_lower_lyr_num
            lb = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
            pl = helpers.plain_batch(lb, ds, dtype)
            got = oracle.run_layers(specs[:k + 1], sd, pl, of)[1:]
            hb = blgnn.BatchData(bp, ds, is_train=False)
            ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(hb.batch_gids))
            got += oracle.run_layers(specs[k + 1:], sd, helpers.plain_batch(hb, ds, dtype, init_x=pl["init_x"],
                                                                           with_pairs=ids), of, first=k + 1)[1:]
        else:
            unique = flags.batch_unique_graphs
            bd = blgnn.BatchData(bp, ds, is_train=False, unique_graphs=unique)
            ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids)) \
                if unique else torch.arange(2 * len(bp)).reshape(-1, 2)
            got = oracle.run_layers(specs, sd, helpers.plain_batch(bd, ds, dtype, with_pairs=ids), of)[1:]
        assert len(got) == len(acts)
        for g, a in zip(got, acts):
            assert max_err(g, a) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(TAGS))
def test_hip_path_matches_fixture(fx, tag, device):
    ds, flags, model, data, specs, acts, bp = _setup(fx, tag, device)
    if flags.higher_level_layers:
        k = flags.last_lower_lyr_num
        preds, _ = bilevel_forward(model, data, bp)
        assert max_err(ds.interaction_combo_nxgraph.init_x, acts[k]) < TOL      # pooled rows == init_x
        got = model.acts[1:]
        ref = acts[k + 1:]
    else:
        bd = blgnn.BatchData(bp, ds, is_train=False, unique_graphs=flags.batch_unique_graphs)
        model.acts = [bd.merge_data['merge'].x]
        for layer in model.layers[:-1]:
            model.acts.append(layer(model.acts[-1], bd, model))
        got, ref = model.acts[1:], acts
    assert len(got) == len(ref)
    for g, a in zip(got, ref):
        assert max_err(g, a) < TOL
