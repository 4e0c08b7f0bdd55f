"""CPU: host logic of the drop-in modules with the test-only fake backend
(tests/fake_backend.py): the modules built from the reference's spec strings produce
the same values as the oracle's interpreter of those strings, which pins parameter
names, layer order, act/BN epilogue order, multi-scale column layout, init_x row
writes and pair-id selection.  Arithmetic parity of the kernels is the GPU suite's job."""

import numpy as np
import pytest
import torch

import blgnn
import oracle
import fake_backend
from blgnn.forward import bilevel_forward, all_molecule_pairs
from helpers import (max_err, build, layer_specs, state_dict_cpu, plain_batch, oracle_flags)


def _run_all(model, bd):
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    return model.acts


@pytest.mark.parametrize("lower,upper", [("gin", "gat"), ("gat", "gcn"), ("gcn", "gin")])
def test_bignn_wiring(monkeypatch, lower, upper):
    fake_backend.install(monkeypatch)
    ds = blgnn.make_synthetic(30, 120, seed=1, device="cpu")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type=lower,
                             higher_level_gnn_type=upper, D_lower=16, D_higher=16, device="cpu")
    model, data = build(flags, ds, seed=3)
    specs = layer_specs(flags)
    of = oracle_flags(flags, ds)
    sd = state_dict_cpu(model)
    k = flags.last_lower_lyr_num
    lower_b = blgnn.BatchData(all_molecule_pairs(ds.gs_map.keys()), ds, is_train=False, ignore_pairs=True)
    pl = plain_batch(lower_b, ds)
    oracle.run_layers(specs[:k + 1], sd, pl, of)
    pairs = ds.pair_gids[:20]
    hb = blgnn.BatchData(pairs, ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(hb.batch_gids))
    acts_ref = oracle.run_layers(specs[k + 1:], sd, plain_batch(hb, ds, init_x=pl["init_x"], with_pairs=ids),
                                 of, first=k + 1)
    preds, _ = bilevel_forward(model, data, pairs)
    assert max_err(ds.interaction_combo_nxgraph.init_x, pl["init_x"]) < 1e-5
    assert preds.shape == (20, 1)
    for got, ref in zip(model.acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < 1e-5


def test_gmn_wiring(monkeypatch):
    fake_backend.install(monkeypatch)
    ds = blgnn.make_synthetic(20, 60, seed=2, device="cpu")
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=16, device="cpu")
    model, data = build(flags, ds, seed=4)
    specs = layer_specs(flags)
    pairs = ds.pair_gids[:8]
    bd = blgnn.BatchData(pairs, ds, is_train=False, unique_graphs=False)
    ids = torch.arange(16).reshape(-1, 2)
    acts_ref = oracle.run_layers(specs, state_dict_cpu(model), plain_batch(bd, ds, with_pairs=ids),
                                 oracle_flags(flags, ds))
    acts = _run_all(model, bd)
    assert len(acts) == len(acts_ref)
    for got, ref in zip(acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < 2e-5       # restructured message: summation order differs


def test_gmn_readout_fusion_is_decided_from_the_layer_list():
    """GMNPropagator._readout_of: the pooled-readout epilogues are taken only for a lower-level-only model driven layer by
    layer over model.layers whose readout is a plain mean / sum; the last propagator alone may drop its node rows"""
    from blgnn.config import FLAGS
    ds = blgnn.make_synthetic(12, 30, shape="drugcombo", seed=3, device="cpu", num_edge_types=2)
    flags = blgnn.make_flags("lower_level_gmn", dataset="drugcombo", lower_level_gnn_type="gin", D_lower=64, device="cpu")
    model, _ = build(flags, ds, seed=4, num_hyper_edge_feat=3)
    layers = list(model.layers)
    props = [k for k, l in enumerate(layers) if type(l).__name__ == "GMNPropagator"]
    assert len(props) == 3
    for k in props:
        model.acts = [None] * (k + 1)                              # the forward loop is about to call layer k
        assert layers[k]._readout_of(model) == ("mean", k == props[-1])
        model.acts = [None] * (k + 2)                              # called out of turn: not the model's own forward
        assert layers[k]._readout_of(model) == (None, False)
    model.acts = [None] * (props[-1] + 1)
    FLAGS.gmn_pooled_tail = False
    assert layers[props[-1]]._readout_of(model) == (None, False)
    FLAGS.gmn_pooled_tail = True
    agg = layers[props[-1] + 1]
    keep = agg.concat_multi_scale
    agg.concat_multi_scale = False                                 # readout of the last layer only: earlier rows are not pooled
    model.acts = [None] * (props[0] + 1)
    assert layers[props[0]]._readout_of(model) == (None, False)
    model.acts = [None] * (props[-1] + 1)
    assert layers[props[-1]]._readout_of(model) == ("mean", True)
    agg.concat_multi_scale = keep
    FLAGS.higher_level_layers = True                               # a bi-level model reads init_x rows: node rows stay
    assert layers[props[-1]]._readout_of(model) == (None, False)
    FLAGS.higher_level_layers = False


def test_drugcombo_meta_wiring(monkeypatch):
    fake_backend.install(monkeypatch)
    ds = blgnn.make_synthetic(25, 100, seed=3, device="cpu", num_edge_types=2)
    flags = blgnn.make_flags("higher_level_gnn", dataset="drugcombo", device="cpu", D_higher=16, d_init=16)
    model, data = build(flags, ds, seed=5, num_hyper_edge_feat=3)
    ds.init_interaction_graph_embds("cpu")
    specs = layer_specs(flags)
    assert specs[1].startswith("MetaLayer:output_dim=16,act=relu,higher_level=True,edge_model=none,"
                               "node_model=gat_multi_edge_aggr")
    bd = blgnn.BatchData(ds.pair_gids[:10], ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    init_x = ds.interaction_combo_nxgraph.init_x
    acts_ref = oracle.run_layers(specs, state_dict_cpu(model),
                                 plain_batch(bd, ds, init_x=init_x, with_pairs=ids), oracle_flags(flags, ds))
    acts = _run_all(model, bd)
    assert acts[-1].shape == (10, 3)
    for got, ref in zip(acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < 1e-5


def test_state_dict_names_match_reference_contract():
    """parameter names a reference checkpoint would carry."""
    ds = blgnn.make_synthetic(10, 20, seed=1, device="cpu")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", device="cpu")
    model, _ = build(flags, ds)
    keys = set(model.state_dict().keys())
    for k in ["layers.0.conv.nn.0.weight", "layers.0.conv.nn.0.bias", "layers.0.conv.nn.2.weight",
              "layers.0.conv.eps", "layers.0.bn.running_mean", "layers.0.bn.num_batches_tracked",
              "layers.7.conv.weight", "layers.7.conv.att", "layers.7.conv.bias",
              "layers.10.mlp_concat.layers.0.weight", "pred_layer.mlp_concat.layers.0.weight",
              "init_layers.0.conv.nn.0.weight", "lower_layers.4.bn.weight",
              "higher_level_layers.1.conv.att"]:
        assert k in keys, k
    sd = model.state_dict()
    assert tuple(sd["layers.0.conv.nn.0.weight"].shape) == (64, 52)      # nn.Linear [out,in]
    assert tuple(sd["layers.7.conv.weight"].shape) == (320, 64)          # PyG [in,out]
    assert tuple(sd["layers.7.conv.att"].shape) == (1, 1, 128)


def test_spec_strings_match_reference_config():
    """src/config.py:297-433 for the BI-GNN model."""
    f = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin")
    assert f.layer_num == 12 and f.last_lower_lyr_num == 5
    assert f.layer_1 == "NodeEmbedding:type=gin,output_dim=64,act=relu,bn=True,normalize=False"
    assert f.layer_5 == "NodeEmbedding:type=gin,input_dim=64,output_dim=64,act=identity,bn=True,normalize=False"
    assert f.layer_6 == "NodeAggregation:style=avg_pool,concat_multi_scale=True,in_dim=64,out_dim=64"
    assert f.layer_7 == "LoadInteractionLayer"
    assert f.layer_8 == ("NodeEmbedding:type=gat,input_dim=320,output_dim=64,act=relu,bn=True,"
                         "higher_level=True,normalize=False")
    assert f.layer_11 == "LinkPredictor:type=mlp_concat,multi_label_pred=False,mlp_dim=64,batch_unique_graphs=True"
    assert f.layer_12 == "Loss:type=BCE"
    g = blgnn.make_flags("lower_level_gmn")
    assert g.batch_size == 128 and not g.batch_unique_graphs and g.layer_num == 9
    assert g.layer_2 == "GMNPropagator:input_dim=64,output_dim=64,act=relu,bn=True,f_node=MLP"
    assert g.layer_8 == "LinkPredictor:type=mlp_concat,multi_label_pred=False,mlp_dim=192,batch_unique_graphs=False"
    h = blgnn.make_flags("higher_level_gnn")
    assert h.layer_num == 6 and h.layer_2 == \
        "NodeEmbedding:type=gat,output_dim=64,act=relu,bn=True,higher_level=True,normalize=False"
    with pytest.raises(ValueError):
        blgnn.layers_factory.create_layers(None, "nope", 0) or blgnn.NodeEmbedding("xyz", 1, 1, "relu", False, False)
    with pytest.raises(ValueError):
        blgnn.layers_util.create_act("gelu")
