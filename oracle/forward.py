"""Layer-level CPU restatement: an interpreter for the reference's layer-spec
strings (src/config.py:297-433, parsed as model/layers_factory.py:14-30 does)
driven by a reference-shaped ``state_dict`` (parameter names: SURVEY §8b).

Test infrastructure only (see oracle/__init__.py); parity unpinned.

``plain`` batch dict (what src/batch.py + src/merged_graph.py hand to the layers,
as plain CPU tensors):
  x [Nn,F], edge_index [2,nnz] int64, batch [Nn] int64, ind_list [(s,e)],
  graph_rows int64[G] | None  -- interaction row gs_map[gid] of each batch graph,
  init_x [N,D] | None         -- interaction init_x before this forward,
  hl_edge_index [2,E] | None, hl_edges [[2,E_k], ...] | None,
  pair_ids int64[B,2] | None  -- LinkPred row ids.
"""

import torch

from . import ops

__all__ = ["parse_spec", "run_layers"]


def parse_spec(spec):
    """model/layers_factory.py:17-25."""
    sp = spec.split(":")
    info = {}
    if len(sp) > 1:
        assert len(sp) == 2
        for kv in sp[1].split(","):
            k = kv.split("=")
            info[k[0]] = "=".join(k[1:])
    return sp[0], info


def _b(s):
    if s == "True":
        return True
    if s == "False":
        return False
    raise RuntimeError("Unknown bool string {}".format(s))


def _lin(sd, prefix):
    return sd[prefix + ".weight"], sd[prefix + ".bias"]


def _bn(sd, prefix):
    return (sd[prefix + ".weight"], sd[prefix + ".bias"],
            sd[prefix + ".running_mean"], sd[prefix + ".running_var"])


def _mlp_params(sd, prefix):
    """MLP state: ``prefix.layers.k.{weight,bias}``, ``prefix.bn.k.*``."""
    lins, k = [], 0
    while prefix + ".layers.%d.weight" % k in sd:
        lins.append(_lin(sd, prefix + ".layers.%d" % k))
        k += 1
    bns, k = [], 0
    while prefix + ".bn.%d.weight" % k in sd:
        bns.append(_bn(sd, prefix + ".bn.%d" % k))
        k += 1
    return lins, (bns or None)


def _node_embedding(x, info, sd, p, plain, flags):
    """model/layers.py:42-63."""
    higher = _b(info["higher_level"]) if info.get("higher_level") else False
    ei = plain["hl_edge_index"] if higher else plain["edge_index"]
    t, act = info["type"], info["act"]
    pw = sd.get(p + ".act.weight")
    if t == "gin":
        x = ops.gin_conv(x, ei, *_lin(sd, p + ".conv.nn.0"), *_lin(sd, p + ".conv.nn.2"),
                         act=act, eps=float(sd.get(p + ".conv.eps", torch.zeros(1))[0]),
                         prelu_weight=pw)
    elif t == "gat":
        x = ops.gat_conv(x, ei, sd[p + ".conv.weight"], sd[p + ".conv.att"],
                         sd[p + ".conv.bias"], flags.get("softmax_group", "src"))
    elif t == "gcn":
        x = ops.gcn_conv(x, ei, sd[p + ".conv.weight"], sd[p + ".conv.bias"],
                         plain.get("edge_weight"))
    else:
        raise ValueError("Unknown node embedding layer type {}".format(t))
    x = ops.apply_act(x, act, pw)
    if _b(info["bn"]):
        if flags.get("train"):
            x = ops.batch_norm_train(x, sd[p + ".bn.weight"], sd[p + ".bn.bias"])
        else:
            x = ops.batch_norm_eval(x, *_bn(sd, p + ".bn"))
    if _b(info["normalize"]):
        x = ops.l2_normalize(x)
    return x


def _aggregation(x, info, sd, p, plain, flags, acts):
    """model/layers_aggregation.py:27-42,66-75."""
    style = info["style"]
    batch, G = plain["batch"], len(plain["ind_list"])
    multi = _b(info["concat_multi_scale"]) if info.get("concat_multi_scale") else False
    if style in ("avg_pool", "sum"):
        if multi:
            sel = acts[1:] if "gmn" not in flags["model"] else \
                [a for i, a in enumerate(acts[1:]) if i % 2 == 1]
        else:
            sel = [x]
        out = ops.node_aggregation(sel, batch, G, style)
    elif style == "deepsets":
        out = ops.deepsets(x, batch, G, _mlp_params(sd, p + ".agg_func.phi")[0],
                           _mlp_params(sd, p + ".agg_func.rho")[0])
    elif style == "gmn_aggr":
        out = ops.gmn_aggregator(x, batch, G,
                                 _mlp_params(sd, p + ".agg_func.weight_func"),
                                 _mlp_params(sd, p + ".agg_func.gate_func"),
                                 _mlp_params(sd, p + ".agg_func.mlp_graph"), train=bool(flags.get("train")))
    else:
        raise NotImplementedError("{} is not implemented".format(style))
    if flags.get("higher_level_layers") and plain.get("graph_rows") is not None:
        if plain.get("init_x") is None:
            plain["init_x"] = torch.zeros(int(flags["num_interaction_nodes"]),
                                          out.shape[1], dtype=out.dtype)
        plain["init_x"][plain["graph_rows"]] = out
    return out


def run_layers(specs, sd, plain, flags, first=0, x0=None):
    """model/model.py:34-62 for one slice of layers.  ``specs`` is the slice of
    spec strings, ``first`` the index of specs[0] in the full layer list (the
    state-dict prefix is ``layers.<index>``).  Returns the ``acts`` list."""
    acts = [plain["x"] if x0 is None else x0]
    for k, spec in enumerate(specs):
        name, info = parse_spec(spec)
        p = "layers.%d" % (first + k)
        x = acts[-1]
        if name == "NodeEmbedding":
            out = _node_embedding(x, info, sd, p, plain, flags)
        elif name == "GMNPropagator":
            if info.get("f_node") == "GRU":
                gru = tuple(sd[p + ".f_node." + k] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                out = ops.gmn_propagator(x, plain["edge_index"], plain["ind_list"],
                                         _mlp_params(sd, p + ".f_messasge")[0], None, None, gru=gru)
            else:
                out = ops.gmn_propagator(
                    x, plain["edge_index"], plain["ind_list"],
                    _mlp_params(sd, p + ".f_messasge")[0],
                    *_mlp_params(sd, p + ".f_node"), train=bool(flags.get("train")))
        elif name in ("NodeAggregation", "NodeAggregationPairs"):
            out = _aggregation(x, info, sd, p, plain, flags, acts)
        elif name == "LoadInteractionLayer":
            out = plain["init_x"]
        elif name == "MetaLayer":
            params, k2 = [], 0
            while p + ".node_model.GNNS.%d.weight" % k2 in sd:
                q = p + ".node_model.GNNS.%d" % k2
                params.append((sd[q + ".weight"], sd.get(q + ".att"), sd[q + ".bias"]))
                k2 += 1
            if params[0][1] is None:                 # NodeModelAggrByEdge('gcn'): model/layers_meta.py:68,74-79
                out = sum(ops.gcn_conv(x, ei, w, b) for ei, (w, _, b) in zip(plain["hl_edges"], params))
            else:
                out = ops.multi_edge_gat(x, plain["hl_edges"], params,
                                         flags.get("softmax_group", "src"))
            out = ops.apply_act(out, info["act"])
        elif name == "LinkPredictor":
            ids = plain["pair_ids"]
            out = ops.link_pred(
                x, ids[:, 0], ids[:, 1],
                _mlp_params(sd, p + ".mlp_concat")[0] if info["type"] == "mlp_concat" else None,
                info["type"],
                _b(info["multi_label_pred"]) if info.get("multi_label_pred") else False)
        elif name == "Loss":
            break
        else:
            raise ValueError("Unknown layer {}".format(name))
        acts.append(out)
    return acts
