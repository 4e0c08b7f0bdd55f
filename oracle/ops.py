"""Op-level CPU restatement (see oracle/__init__.py: test infrastructure, parity unpinned).

Every function cites the reference call site (path:line under /root/reference)
whose arithmetic it restates.  All functions are dtype-agnostic: feed fp64 to get
a higher-precision check value, fp32 to mimic the reference CPU path.

Message-passing convention of PyG 1.1.2 (``flow='source_to_target'``):
``x_j = x[edge_index[0]]`` (source), ``x_i = x[edge_index[1]]`` (target),
messages are summed at ``edge_index[1]``.
"""

import torch
import torch.nn.functional as F

__all__ = [
    "to_undirected", "coalesce_max", "remove_self_loops", "add_self_loops",
    "scatter_add", "scatter_mean", "segment_softmax", "apply_act",
    "batch_norm_eval", "l2_normalize", "gin_conv", "gat_conv", "gcn_conv",
    "gcn_norm", "mlp", "gmn_match", "gmn_message_sum", "gmn_propagator",
    "node_aggregation", "deepsets", "gmn_aggregator", "merge_graphs",
    "link_pred", "multi_edge_gat",
]


# --------------------------------------------------------------------------
# graph utilities (third-party: torch_geometric.utils / torch_sparse.coalesce)
# --------------------------------------------------------------------------

def _coalesce_index(edge_index, num_nodes):
    key = edge_index[0] * num_nodes + edge_index[1]
    uniq, inv = torch.unique(key, sorted=True, return_inverse=True)
    out = torch.stack([uniq // num_nodes, uniq % num_nodes], dim=0)
    return out, inv


def to_undirected(edge_index, num_nodes):
    """model/layers_util.py:166, src/batch.py:172: both directions, sorted by
    (row, col), duplicates removed (torch_sparse.coalesce)."""
    if edge_index.numel() == 0:
        return edge_index.reshape(2, 0).long()
    row, col = edge_index
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])], dim=0)
    return _coalesce_index(both, num_nodes)[0]


def coalesce_max(edge_index, edge_attr, num_nodes):
    """model/layers_util.py:156-164: symmetrise edges + attrs, coalesce(op='max')."""
    row, col = edge_index
    both = torch.stack([torch.cat([row, col]), torch.cat([col, row])], dim=0)
    attr = torch.cat([edge_attr, edge_attr], dim=0)
    out, inv = _coalesce_index(both, num_nodes)
    shape = (out.shape[1],) + tuple(attr.shape[1:])
    red = torch.full(shape, float("-inf"), dtype=attr.dtype)
    idx = inv.view(-1, *([1] * (attr.dim() - 1))).expand_as(attr)
    red = red.scatter_reduce(0, idx, attr, reduce="amax", include_self=True)
    return out, red


def remove_self_loops(edge_index, edge_attr=None):
    keep = edge_index[0] != edge_index[1]
    ei = edge_index[:, keep]
    return ei, (None if edge_attr is None else edge_attr[keep])


def add_self_loops(edge_index, num_nodes):
    """PyG 1.1.2 appends the N loops AFTER the existing edges."""
    loop = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1)


# --------------------------------------------------------------------------
# torch_scatter 1.1.2
# --------------------------------------------------------------------------

def scatter_add(src, index, dim_size=None):
    """model/layers_gmn.py:42, model/layers_aggregation.py:19,94."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def scatter_mean(src, index, dim_size=None):
    """model/layers_aggregation.py:17,54: sum / count.clamp(min=1)."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    s = scatter_add(src, index, dim_size)
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add_(
        0, index, torch.ones(index.shape[0], dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return s / cnt.view(-1, *([1] * (src.dim() - 1)))


def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax 1.1.2: exp(a - segmax) / (segsum + 1e-16)."""
    mx = torch.full((num_nodes,), float("-inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros(num_nodes, dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


# --------------------------------------------------------------------------
# dense helpers
# --------------------------------------------------------------------------

def apply_act(x, name, prelu_weight=None):
    """model/layers_util.py:60-76."""
    if name == "relu":
        return torch.relu(x)
    if name == "prelu":
        return F.prelu(x, prelu_weight)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "identity":
        return x
    raise ValueError("Unknown activation function {}".format(name))


def batch_norm_eval(x, weight, bias, running_mean, running_var, eps=1e-5):
    """model/layers.py:56-57 in eval mode (forward-parity target)."""
    return (x - running_mean) / torch.sqrt(running_var + eps) * weight + bias


def batch_norm_train(x, weight, bias, eps=1e-5):
    """model/layers.py:56-57 in train mode (torch.nn.BatchNorm1d: batch mean, BIASED batch variance);
    differentiable, used by the gradient-parity tests of the training step (src/train.py:177-182)."""
    mean = x.mean(dim=0)
    var = ((x - mean) ** 2).mean(dim=0)
    return (x - mean) / torch.sqrt(var + eps) * weight + bias


def l2_normalize(x, eps=1e-12):
    """torch.nn.functional.normalize(p=2, dim=1): model/layers.py:60-61,
    model/layers_link_pred.py:44, model/layers_gmn.py:75-76."""
    return x / x.norm(2, dim=1, keepdim=True).clamp_min(eps)


def mlp(x, linears, act="relu", bns=None, train=False):
    """model/layers_util.py:44-57.  ``linears`` = [(W[out,in], b[out]), ...];
    every layer but the last is Linear -> [BN] -> act; the last is Linear only.
    ``bns`` = [(weight, bias, mean, var), ...] for the hidden layers or None; ``train``: batch statistics."""
    for i, (w, b) in enumerate(linears):
        x = x @ w.t() + b
        if i != len(linears) - 1:
            if bns:
                x = batch_norm_train(x, bns[i][0], bns[i][1]) if train else batch_norm_eval(x, *bns[i])
            x = apply_act(x, act)
    return x


# --------------------------------------------------------------------------
# PyG 1.1.2 convolutions (call sites: model/layers.py:23-33,51-54)
# --------------------------------------------------------------------------

def gin_conv(x, edge_index, w1, b1, w2, b2, act, eps=0.0, prelu_weight=None):
    """GINConv(nn=Linear->act->Linear): model/layers.py:26-31,54.
    out = nn((1+eps) x + sum_{j->i, j!=i} x_j); the SAME act module is reused
    after the conv by the caller (model/layers.py:55)."""
    ei, _ = remove_self_loops(edge_index)
    agg = scatter_add(x[ei[0]], ei[1], x.shape[0])
    h = (1.0 + eps) * x + agg
    h = apply_act(h @ w1.t() + b1, act, prelu_weight)
    return h @ w2.t() + b2


def gat_conv(x, edge_index, weight, att, bias, softmax_group="src",
             negative_slope=0.2):
    """GATConv(heads=1, concat): model/layers.py:33,54; model/layers_meta.py:70,78.
    weight [in,out] (x @ W), att [1,1,2*out] (first half multiplies the TARGET
    row x_i), bias [out].  ``softmax_group``: 'src' = group by edge_index[0]
    (the 1.1.x call ``softmax(alpha, edge_index[0], num_nodes)``), 'dst' = group
    by edge_index[1] (textbook GAT).  SURVEY §8(c)(5)."""
    n = x.shape[0]
    ei, _ = remove_self_loops(edge_index)
    ei = add_self_loops(ei, n)
    h = x @ weight
    d = h.shape[1]
    a = att.reshape(-1)
    src, dst = ei[0], ei[1]
    alpha = (h[dst] * a[:d]).sum(-1) + (h[src] * a[d:]).sum(-1)
    alpha = F.leaky_relu(alpha, negative_slope)
    group = src if softmax_group == "src" else dst
    alpha = segment_softmax(alpha, group, n)
    out = scatter_add(h[src] * alpha.view(-1, 1), dst, n)
    return out + bias


def gcn_norm(edge_index, num_nodes, edge_weight=None, dtype=torch.float32):
    """GCNConv.norm of PyG 1.1.2; the reference's copy at
    model/layers_meta.py:128-148 documents the formula (deg by edge_index[0])."""
    if edge_weight is None:
        edge_weight = torch.ones(edge_index.shape[1], dtype=dtype)
    edge_weight = edge_weight.view(-1)
    ei, ew = remove_self_loops(edge_index, edge_weight)
    ei = add_self_loops(ei, num_nodes)
    ew = torch.cat([ew, torch.ones(num_nodes, dtype=ew.dtype)])
    row, col = ei
    deg = scatter_add(ew, row, num_nodes)
    dis = deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    return ei, dis[row] * ew * dis[col]


def gcn_conv(x, edge_index, weight, bias, edge_weight=None):
    """GCNConv: model/layers.py:23,52."""
    n = x.shape[0]
    h = x @ weight
    ei, norm = gcn_norm(edge_index, n, edge_weight, h.dtype)
    out = scatter_add(h[ei[0]] * norm.view(-1, 1), ei[1], n)
    return out + bias


def multi_edge_gat(x, edge_indices, params, softmax_group="src"):
    """NodeModelAggrByEdge('gat'): model/layers_meta.py:74-79.
    ``params`` = [(weight, att, bias), ...] one per edge type."""
    out = torch.zeros(x.shape[0], params[0][0].shape[1], dtype=x.dtype)
    for ei, (w, a, b) in zip(edge_indices, params):
        out = out + gat_conv(x, ei, w, a, b, softmax_group)
    return out


# --------------------------------------------------------------------------
# GMN (model/layers_gmn.py) -- plain torch in the reference, restated 1:1
# --------------------------------------------------------------------------

def _match_one(g1x, g2x):
    """model/layers_gmn.py:74-84."""
    g1n = l2_normalize(g1x)
    g2n = l2_normalize(g2x)
    a1 = torch.softmax(g1n @ g2n.t(), dim=1)
    return g1x - a1 @ g2x


def gmn_match(x, ind_list):
    """model/layers_gmn.py:55-72: consecutive graphs (2k, 2k+1) form a pair."""
    outs = []
    for i in range(0, len(ind_list), 2):
        a0, a1 = ind_list[i]
        b0, b1 = ind_list[i + 1]
        g1x, g2x = x[a0:a1], x[b0:b1]
        outs.append(_match_one(g1x, g2x))
        outs.append(_match_one(g2x, g1x))
    return torch.cat(outs, dim=0).view(x.shape[0], -1)


def gmn_message_sum(x, edge_index, msg_linears):
    """model/layers_gmn.py:38-42: m = f_message(cat(x[row], x[col]));
    m_sum = scatter_add(m, row)."""
    row, col = edge_index
    m = mlp(torch.cat((x[row], x[col]), dim=1), msg_linears, "relu")
    return scatter_add(m, row, x.shape[0])


def gru_cell(inp, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRUCell restated: gate order r | z | n."""
    gi = inp @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    D = h.shape[1]
    r = torch.sigmoid(gi[:, :D] + gh[:, :D])
    z = torch.sigmoid(gi[:, D:2 * D] + gh[:, D:2 * D])
    c = torch.tanh(gi[:, 2 * D:] + r * gh[:, 2 * D:])
    return (1 - z) * c + z * h


def gmn_propagator(x, edge_index, ind_list, msg_linears, node_linears, node_bns, train=False, gru=None):
    """model/layers_gmn.py:34-53: f_node='MLP' (bn=True), or 'GRU' when ``gru`` = (w_ih, w_hh, b_ih, b_hh)."""
    m_sum = gmn_message_sum(x, edge_index, msg_linears)
    u_sum = gmn_match(x, ind_list)
    if gru is not None:
        return gru_cell(torch.cat((m_sum, u_sum), dim=1), x, *gru)
    return mlp(torch.cat((x, m_sum, u_sum), dim=1), node_linears, "relu", node_bns, train=train)


# --------------------------------------------------------------------------
# readout (model/layers_aggregation.py)
# --------------------------------------------------------------------------

def deepsets(x, batch, num_graphs, phi, rho):
    """model/layers_aggregation.py:45-56."""
    return mlp(scatter_mean(mlp(x, phi), batch, num_graphs), rho)


def gmn_aggregator(x, batch, num_graphs, weight_f, gate_f, graph_f, train=False):
    """model/layers_aggregation.py:78-95; each arg = (linears, bns); ``train``: batch statistics in the MLPs' BN."""
    wx = mlp(x, weight_f[0], "relu", weight_f[1], train)
    gx = torch.sigmoid(mlp(x, gate_f[0], "relu", gate_f[1], train))
    ge = scatter_add(gx * wx, batch, num_graphs)
    return mlp(ge, graph_f[0], "relu", graph_f[1], train)


def node_aggregation(acts, batch, num_graphs, style="avg_pool"):
    """model/layers_aggregation.py:27-42: ``acts`` is the list of activations to
    pool (one tensor for single-scale, the per-layer list for multi-scale);
    results are concatenated along dim 1."""
    if style == "avg_pool":
        f = scatter_mean
    elif style == "sum":
        f = scatter_add
    else:
        raise NotImplementedError("{} is not implemented".format(style))
    return torch.cat([f(a, batch, num_graphs) for a in acts], dim=1)


# --------------------------------------------------------------------------
# batch construction (src/merged_graph.py:27-96)
# --------------------------------------------------------------------------

def merge_graphs(graphs):
    """Block-diagonal merge of [(x [n_i,F], edge_index [2,e_i]), ...]."""
    xs, eis, batch, ind_list, edge_ind_list = [], [], [], [], []
    cn = ce = 0
    for i, (x, ei) in enumerate(graphs):
        n, e = x.shape[0], ei.shape[1]
        xs.append(x)
        eis.append(ei + cn)
        batch.append(torch.full((n,), i, dtype=torch.long))
        ind_list.append((cn, cn + n))
        edge_ind_list.append((ce, ce + e))
        cn += n
        ce += e
    return {
        "x": torch.cat(xs, dim=0),
        "edge_index": torch.cat(eis, dim=1),
        "batch": torch.cat(batch),
        "ind_list": ind_list,
        "edge_ind_list": edge_ind_list,
    }


# --------------------------------------------------------------------------
# pair scorer (model/layers_link_pred.py:43-67)
# --------------------------------------------------------------------------

def link_pred(ins, ids1, ids2, linears=None, type="mlp_concat",
              multi_label_pred=False):
    z = l2_normalize(ins)
    g1x, g2x = z[ids1], z[ids2]
    if type == "mlp_concat":
        p = mlp(torch.cat((g1x, g2x), dim=1), linears, "relu")
        return p if multi_label_pred else torch.sigmoid(p)
    if type == "dot_product":
        return torch.sigmoid((g1x * g2x).sum(dim=1))
    raise NotImplementedError
