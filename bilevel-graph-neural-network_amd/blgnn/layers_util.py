"""MLP / activation / graph-conversion helpers of the hot path.

Drop-in for the parts of the reference's ``model/layers_util.py`` the path uses:
``MLP`` (:12-57), ``create_act`` (:60-76), ``convert_nx_to_pyg_graph`` (:100-135)
and ``create_edge_index`` (:148-167).  Dense math runs on the fp32 matrix cores
through ``blgnn_linear``; there is no torch.mm on the forward path.
"""

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import FLAGS
from .graph import GraphData, to_undirected_host

__all__ = ["MLP", "create_act", "Identity", "act_name", "DerivedCache", "bn_affine",
           "convert_nx_to_pyg_graph", "create_edge_index"]


class Identity(nn.Module):
    def forward(self, x):
        return x


def create_act(act, num_parameters=None):
    """Same names and error as model/layers_util.py:60-76.  The returned modules
    carry the parameters (PReLU) -- the arithmetic itself is fused into the
    producing kernel's epilogue."""
    if act == "relu":
        return nn.ReLU()
    if act == "prelu":
        return nn.PReLU(num_parameters)
    if act == "sigmoid":
        return nn.Sigmoid()
    if act == "tanh":
        return nn.Tanh()
    if act == "identity":
        return Identity()
    raise ValueError("Unknown activation function {}".format(act))


def act_name(module):
    for cls, name in ((nn.ReLU, "relu"), (nn.PReLU, "prelu"), (nn.Sigmoid, "sigmoid"),
                      (nn.Tanh, "tanh"), (Identity, "identity")):
        if isinstance(module, cls):
            return name
    raise ValueError("Unknown activation module {}".format(type(module)))


class DerivedCache(object):
    """Tensors derived from parameters (transposed nn.Linear weights, folded eval-BN
    affine, expanded PReLU slopes), recomputed only when a source tensor changes
    (``_version`` bump, re-allocation or device move)."""

    def __init__(self):
        self._store = {}

    def get(self, key, fn, *srcs):
        stamp = tuple((s.data_ptr(), s._version, str(s.device)) for s in srcs)
        hit = self._store.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, fn(*srcs))
            self._store[key] = hit
        return hit[1]


def bn_affine(cache, key, bn):
    """eval-mode BatchNorm1d as a per-channel affine (scale, shift) for the epilogue."""
    if not bn:
        return None, None
    if bn.training:
        raise NotImplementedError(
            "blgnn: train-mode BatchNorm couples all rows of a batch and is outside the "
            "forward-parity scope (SURVEY section 7); call model.eval()")

    def fold(w, b, mean, var):
        scale = w / torch.sqrt(var + bn.eps)
        return torch.stack([scale, b - mean * scale]).contiguous()

    ss = cache.get(key, fold, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    return ss[0], ss[1]


def prelu_slopes(cache, key, act, dim):
    if not isinstance(act, nn.PReLU):
        return None
    return cache.get(key, lambda w: w.expand(dim).contiguous(), act.weight)


class MLP(nn.Module):
    """model/layers_util.py:12-57: Linear -> [BN] -> act per hidden layer, final
    Linear bare; Xavier-uniform(gain=sqrt 2) weights.  Parameter names match the
    reference (``layers.k.{weight,bias}``, ``bn.k.*``)."""

    def __init__(self, input_dim, output_dim, activation_type="relu", num_hidden_lyr=2,
                 hidden_channels=None, bn=False):
        super().__init__()
        self.out_dim = output_dim
        if not hidden_channels:
            hidden_channels = [input_dim for _ in range(num_hidden_lyr)]
        elif len(hidden_channels) != num_hidden_lyr:
            raise ValueError("number of hidden layers should be the same as the lengh of hidden_channels")
        self.layer_channels = [input_dim] + hidden_channels + [output_dim]
        self.activation = create_act(activation_type)
        lins = []
        for a, b in zip(self.layer_channels[:-1], self.layer_channels[1:]):
            lin = nn.Linear(a, b)
            nn.init.xavier_uniform_(lin.weight, gain=nn.init.calculate_gain("relu"))
            lins.append(lin)
        self.layers = nn.ModuleList(lins)
        self.bn = bn
        if self.bn:
            self.bn = nn.ModuleList([nn.BatchNorm1d(c) for c in self.layer_channels[1:-1]])
        self._cache = DerivedCache()

    def weight_kn(self, i):
        return self._cache.get(("wt", i), lambda w: w.t().contiguous(), self.layers[i].weight)

    def _folded(self, i):
        """Hidden layer i with its eval BatchNorm folded in.  The reference order is
        Linear -> BN -> act (layers_util.py:51-54) while the GEMM epilogue is
        bias -> act -> affine, so the per-channel affine moves into the weights:
        W' = W^T * scale, b' = b * scale + shift (one extra rounding per weight)."""
        bn = self.bn[i]
        if bn.training:
            raise NotImplementedError("blgnn: train-mode BatchNorm is outside the forward-parity scope")

        def fold(w, b, g, beta, mean, var):
            scale = g / torch.sqrt(var + bn.eps)
            return (w.t() * scale.view(1, -1)).contiguous(), (b * scale + beta - mean * scale).contiguous()

        lin = self.layers[i]
        return self._cache.get(("fold", i), fold, lin.weight, lin.bias, bn.weight, bn.bias,
                               bn.running_mean, bn.running_var)

    def forward(self, x, out=None, first_rowvec=None, final_act="identity", row_amax=None):
        """``out``: optional destination for the last layer (may be a column slice).
        ``final_act``: activation fused after the last Linear (the caller's sigmoid).
        ``first_rowvec`` = (rowvec, rowvec_w): rank-1 term added inside the first
        layer's GEMM (used by the restructured GMN message).  ``row_amax``: row magnitudes
        of ``x`` if the caller has them; between layers they come out of the producing GEMM's
        epilogue (f16x3 path), so only the first layer may need a pass over its input."""
        act = act_name(self.activation)
        last = len(self.layers) - 1
        if self.training and torch.is_grad_enabled() and x.requires_grad:
            # train mode: Linear(+act) autograd nodes on libblgnn kernels (blgnn/autograd.py)
            from . import autograd as ag
            if first_rowvec is not None or out is not None:
                raise NotImplementedError("blgnn: train mode takes neither a rank-1 term nor a destination slice")
            for i, lin in enumerate(self.layers):
                if i == last:
                    x = ag.linear(x, lin.weight, lin.bias, final_act)
                elif self.bn:                                    # Linear -> BN -> act (layers_util.py:51-54)
                    x = ag.batch_norm_train(ag.linear(x, lin.weight, lin.bias, "identity"), self.bn[i], act=act)
                else:
                    x = ag.linear(x, lin.weight, lin.bias, act)
            return x
        amax = row_amax
        for i, lin in enumerate(self.layers):
            kw = {}
            if i == 0 and first_rowvec is not None:
                kw["rowvec"], kw["rowvec_w"] = first_rowvec
            if i == last:
                w, b, a = self.weight_kn(i), lin.bias, final_act
                kw["out"] = out
            elif self.bn:
                w, b = self._folded(i)
                a = act
            else:
                w, b, a = self.weight_kn(i), lin.bias, act
            nxt = None
            if i < last and ops.linear_writes_amax(x, w):
                nxt = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
            x = ops.linear(x, w, b, act=a, row_amax=amax, amax_out=nxt, **kw)
            amax = nxt
        return x


def _edges_of(g):
    """Edge list of an interaction graph: ours (``.edges`` ndarray) or networkx."""
    e = g.edges
    if isinstance(e, np.ndarray):
        return e
    return np.asarray(sorted(list(e)), dtype=np.int64).reshape(-1, 2)


def create_edge_index(g, edge_attr_x=None):
    """model/layers_util.py:148-167 without edge attributes (the configs force
    ``use_interaction_edge_attrs=False``: src/config.py:77-83)."""
    if edge_attr_x is not None:
        raise NotImplementedError("interaction edge attributes are outside the hot path")
    ei = to_undirected_host(_edges_of(g), g.number_of_nodes())
    return torch.as_tensor(ei).to(FLAGS.device), None


def convert_nx_to_pyg_graph(g):
    """model/layers_util.py:100-135: graph object with ``init_x`` (+ edges) ->
    GraphData(x, edge_index).  Accepts blgnn.graph.InteractionGraph or nx.Graph."""
    if hasattr(g, "graph_data"):
        return g.graph_data(FLAGS.device)
    import networkx as nx
    if type(g) is not nx.Graph:
        raise ValueError("Input graphs must be undirected nx.Graph, NOT {}".format(type(g)))
    cache = getattr(g, "_blgnn_data", None)
    if cache is None or cache[0] != g.number_of_edges():
        ei, _ = create_edge_index(g)
        cache = (g.number_of_edges(), GraphData(edge_index=ei, num_nodes=g.number_of_nodes()))
        cache[1].symmetric_by_construction = True                # create_edge_index: to_undirected
        g._blgnn_data = cache
    data = cache[1]
    x = g.init_x
    if not isinstance(x, torch.Tensor):
        x = torch.tensor(np.asarray(x), dtype=torch.float32, device=FLAGS.device)
    data.x = x
    return data
