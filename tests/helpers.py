"""Shared test plumbing: build a product Model, export its weights to the oracle,
calibrate eval-BatchNorm statistics so activations stay O(1) (the 1e-5 absolute
parity bar is only meaningful at that scale), and convert a BatchData into the
oracle's plain-tensor batch."""

import copy

import numpy as np
import torch

import blgnn
import oracle
from blgnn.config import set_flags
from blgnn.forward import all_molecule_pairs

TOL = 1e-5   # north star: fp32 max|delta| < 1e-5 vs the CPU oracle, activations O(1)

# every comparison of a test run: test id -> list of {abs, scaled, ref_max, n}; tests/conftest.py writes it
# to gpurun_out/parity_errors.json at the end of the session (the measured errors, not just pass / fail)
ERROR_LOG = {}


def _log(abs_err, scaled, ref_max, n):
    import os
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    ERROR_LOG.setdefault(test, []).append({"abs": abs_err, "scaled": scaled, "ref_max": ref_max, "n": n})


def abs_err(got, ref):
    """absolute max|got - ref| (the north-star quantity), recorded"""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    if not ref.numel():
        return 0.0
    a = float((got - ref).abs().max())
    rm = float(ref.abs().max())
    _log(a, a / max(1.0, rm), rm, int(ref.numel()))
    return a


def max_err(got, ref):
    """max|got - ref| / max(1, max|ref|): absolute below magnitude 1, relative to the tensor's largest
    entry above it (fp32 carries 24 bits: the rounding of ANY fp32 evaluation scales with the magnitude)"""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    if not ref.numel():
        return 0.0
    a = float((got - ref).abs().max())
    rm = float(ref.abs().max())
    _log(a, a / max(1.0, rm), rm, int(ref.numel()))
    return a / max(1.0, rm)


def compare3(got, ref64, cpu32):
    """One comparison with the fp32 CPU oracle's own error beside the GPU's: ``got`` (GPU) and ``cpu32`` (the oracle
    evaluated in fp32 on the CPU -- the arithmetic the reference itself runs) against ``ref64`` (the oracle in fp64).
    -> (gpu_scaled, cpu32_scaled); both are recorded in the parity log (keys ``cpu32_abs`` / ``cpu32_scaled``)."""
    got = got.detach().cpu().double()
    ref = ref64.detach().cpu().double()
    c32 = cpu32.detach().cpu().double()
    if not ref.numel():
        return 0.0, 0.0
    rm = float(ref.abs().max())
    a = float((got - ref).abs().max())
    c = float((c32 - ref).abs().max())
    _log(a, a / max(1.0, rm), rm, int(ref.numel()))
    import os
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    ERROR_LOG[test][-1].update({"cpu32_abs": c, "cpu32_scaled": c / max(1.0, rm)})
    return a / max(1.0, rm), c / max(1.0, rm)


def layer_specs(flags):
    return [getattr(flags, "layer_%d" % i) for i in range(1, flags.layer_num + 1)]


def build(flags, dataset, seed=0, num_hyper_edge_feat=0):
    set_flags(flags)
    torch.manual_seed(seed)
    data = blgnn.TrainData(dataset, device=flags.device, num_hyper_edge_feat=num_hyper_edge_feat)
    if dataset.interaction_num_node_feat is None:
        dataset.init_interaction_graph_feats(flags.init_embds, flags.device, flags.d_init, seed=seed)
    model = blgnn.Model(data)
    # non-trivial BN affine + biases so that every epilogue term is exercised
    g = torch.Generator().manual_seed(seed + 1)

    def rnd(p, centre, width):
        with torch.no_grad():
            p.copy_(centre + width * (torch.rand(p.shape, generator=g) - 0.5))

    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            rnd(mod.weight, 1.0, 0.2)
            rnd(mod.bias, 0.0, 0.1)
            rnd(mod.running_mean, 0.0, 0.2)
            rnd(mod.running_var, 1.0, 0.2)
        elif isinstance(mod, (torch.nn.Linear, blgnn.layers.GATConv, blgnn.layers.GCNConv)):
            rnd(mod.bias, 0.0, 0.1)
    model.to(flags.device)
    model.eval()
    return model, data


def state_dict_cpu(model, dtype=torch.float32):
    sd = {}
    for k, v in model.state_dict().items():
        v = v.detach().cpu()
        sd[k] = v.to(dtype) if v.is_floating_point() else v
    return sd


def plain_batch(batch_data, dataset, dtype=torch.float32, init_x=None, with_pairs=None):
    m = batch_data.merge_data['merge']
    plain = {
        "x": m.x.detach().cpu().to(dtype),
        "edge_index": m.edge_index.detach().cpu(),
        "batch": m.batch.detach().cpu(),
        "ind_list": batch_data.merge_data['ind_list'],
        "graph_rows": torch.as_tensor([dataset.gs_map[g] for g in batch_data._graph_gids]),
        "init_x": None if init_x is None else init_x.detach().cpu().to(dtype),
        "hl_edge_index": torch.as_tensor(
            blgnn.graph.to_undirected_host(dataset.interaction_combo_nxgraph.edges, dataset.num_graphs())),
        "hl_edges": None,
        "pair_ids": with_pairs,
    }
    if dataset.interaction_nxgraphs:
        plain["hl_edges"] = [torch.as_tensor(blgnn.graph.to_undirected_host(g.edges, dataset.num_graphs()))
                             for g in dataset.interaction_nxgraphs.values()]
    return plain


def oracle_flags(flags, dataset):
    return {"model": flags.model, "higher_level_layers": flags.higher_level_layers,
            "softmax_group": flags.gat_softmax_group, "num_interaction_nodes": dataset.num_graphs()}


def calibrate_bn(model, specs, plain, oflags, first=0, x0=None):
    """Set every NodeEmbedding BatchNorm's running stats to the statistics of its own
    input on ``plain`` (as a trained model's would be), walking the slice in order."""
    sd = state_dict_cpu(model)
    for k, spec in enumerate(specs):
        name, info = oracle.parse_spec(spec)
        if name != "NodeEmbedding" or info.get("bn") != "True":
            continue
        p = "layers.%d.bn" % (first + k)
        sd[p + ".weight"] = torch.ones_like(sd[p + ".weight"])
        sd[p + ".bias"] = torch.zeros_like(sd[p + ".bias"])
        sd[p + ".running_mean"] = torch.zeros_like(sd[p + ".running_mean"])
        sd[p + ".running_var"] = torch.ones_like(sd[p + ".running_var"]) - 1e-5
        acts = oracle.run_layers(specs[:k + 1], sd, copy.copy(plain), oflags, first=first, x0=x0)
        pre = acts[-1]
        bn = model.layers[first + k].bn
        with torch.no_grad():
            bn.running_mean.copy_(pre.mean(0).to(bn.running_mean.device))
            bn.running_var.copy_(pre.var(0, unbiased=False).clamp_min(0.5).to(bn.running_var.device))
        sd = state_dict_cpu(model)
    return sd
