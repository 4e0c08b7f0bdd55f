"""``NodeEmbedding`` and ``Loss`` -- drop-ins for the reference's model/layers.py.

Same constructor, same ``forward(ins, batch_data, model)``, same parameter names
(GIN ``conv.nn.0/2.{weight,bias}`` + buffer ``conv.eps``; GAT ``conv.{weight,att,
bias}``; GCN ``conv.{weight,bias}``; ``bn.*``; PReLU ``act.weight``), so a
reference ``state_dict`` loads unchanged.  The forward is libblgnn kernels only:
CSR gather/segment-reduce + fp32-MFMA transform with act and eval-BatchNorm fused
into the epilogue.  Forward-only: the reference's ``act -> bn`` order and its GIN
"shared act module" quirk (layers.py:26,29,55) are preserved.
"""

import math

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .config import FLAGS
from .layers_util import create_act, act_name, DerivedCache, bn_affine, prelu_slopes

__all__ = ["NodeEmbedding", "Loss", "GINConv", "GATConv", "GCNConv"]


def _glorot(t):
    bound = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-bound, bound)


class GINConv(nn.Module):
    """Parameter holder shaped like torch_geometric.nn.GINConv(nn) 1.1.2."""

    def __init__(self, mlps, eps=0.0):
        super().__init__()
        self.nn = mlps
        self.register_buffer("eps", torch.tensor([float(eps)]))


class GATConv(nn.Module):
    """Parameter holder shaped like torch_geometric.nn.GATConv(in, out) 1.1.2
    (heads=1, concat=True, negative_slope=0.2, dropout=0)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.negative_slope = 0.2
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.att = nn.Parameter(torch.empty(1, 1, 2 * out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.weight)
        _glorot(self.att)


class GCNConv(nn.Module):
    """Parameter holder shaped like torch_geometric.nn.GCNConv(in, out) 1.1.2."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        _glorot(self.weight)


class ReadoutPlan(object):
    """Node -> graph readout fused into the layers that produce the pooled activations (north_star: "with the
    node readout in layers_aggregation.py fused in").  Built by the first lower GIN layer of a forward when the
    layers that follow it in ``model.layers`` are GIN layers of one width and then a segment-pool readout
    (model/layers_aggregation.py:32-41): every pooled layer adds its rows into its column slice of ``dest`` from
    the kernel epilogue (or, when its kernel has no such epilogue, by one segment-reduce launch), and the readout
    layer finds its result already there."""

    def __init__(self, batch_data, agg, convs, pooled_convs, seg, row_graph, dest):
        self.batch, self.agg, self.convs = batch_data, agg, convs
        self.seg, self.row_graph, self.dest = seg, row_graph, dest
        n = convs[0].out_dim
        self.cols = {id(c): i * n for i, c in enumerate(pooled_convs)}
        self.kind = agg.agg_func.kind
        self.done = set()

    def pool_for(self, layer):
        c = self.cols.get(id(layer))
        if c is None:
            return None
        return (self.row_graph, self.seg, self.dest[:, c:c + layer.out_dim], self.kind)

    def complete(self):
        return len(self.done) == len(self.cols)


def _build_readout_plan(first, model, batch_data):
    from .layers_aggregation import NodeAggregationPairs, _SegmentPool
    if not getattr(FLAGS, "fuse_readout", True) or getattr(FLAGS, "feature_transform", "f16x3") != "f16x3":
        return None
    if 'gmn' in FLAGS.model or first.out_dim not in (128, 256):
        return None
    layers = list(model.layers)
    i = next((k for k, l in enumerate(layers) if l is first), None)
    if i is None:
        return None
    convs, j = [], i
    while j < len(layers) and isinstance(layers[j], NodeEmbedding):
        l = layers[j]
        if l.type != 'gin' or l.higher_level or l.normalize or l.out_dim != first.out_dim:
            return None
        convs.append(l)
        j += 1
    if j >= len(layers) or not isinstance(layers[j], NodeAggregationPairs):
        return None
    agg = layers[j]
    if not isinstance(agg.agg_func, _SegmentPool):
        return None
    pooled = convs if agg.concat_multi_scale else convs[-1:]
    if len(pooled) > 8:
        return None
    g = batch_data.merge_data['merge']
    seg = g.segments()
    G, width = seg.numel() - 1, first.out_dim * len(pooled)
    dest = None
    if getattr(model, "_readout_direct", False) and hasattr(agg, "readout_destination"):
        dest = agg.readout_destination(batch_data, width)          # the init_x rows of the batch graphs
    if dest is None:
        dest = torch.empty((G, width), dtype=torch.float32, device=g.x.device)
    dest.zero_()
    return ReadoutPlan(batch_data, agg, convs, pooled, seg, g.row_graph(), dest)


class NodeEmbedding(nn.Module):
    def __init__(self, type, in_dim, out_dim, act, bn, normalize, higher_level=False,
                 use_edge_attr=None):
        super(NodeEmbedding, self).__init__()
        self.normalize = normalize
        self.type = type
        self.out_dim = out_dim
        self.higher_level = higher_level
        assert higher_level if use_edge_attr and use_edge_attr != 'none' else True
        self.use_edge_attr = use_edge_attr
        if type == 'gcn':
            self.conv = GCNConv(in_dim, out_dim)
            self.act = create_act(act, out_dim)
        elif type == 'gin':
            self.act = create_act(act, out_dim)
            self.conv = GINConv(nn.Sequential(
                nn.Linear(in_dim, out_dim), self.act, nn.Linear(out_dim, out_dim)))
        elif type == 'gat':
            self.conv = GATConv(in_dim, out_dim)
            self.act = create_act(act, out_dim)
        else:
            raise ValueError('Unknown node embedding layer type {}'.format(type))
        self.bn = bn
        if self.bn:
            self.bn = torch.nn.BatchNorm1d(out_dim)
        self._cache = DerivedCache()

    def _graph(self, batch_data):
        if self.higher_level:
            return batch_data.merge_higher_level['merge']
        return batch_data.merge_data['merge']

    def _shared_molecules(self, ins, batch_data, model, pool):
        """``batch_data.shared_molecules()`` when this call is the FIRST layer of an eval forward over a pair batch
        (its input is the merged atom features, nothing pair-dependent yet), the duplication is worth a replication
        pass (FLAGS.gmn_shared_prefix, default on; < 60 % distinct rows) and no readout hangs on this layer."""
        if not getattr(FLAGS, "gmn_shared_prefix", True) or pool is not None or not hasattr(batch_data, "shared_molecules"):
            return None
        if model.acts is None or len(model.acts) != 1 or ins is not self._graph(batch_data).x:
            return None
        sm = batch_data.shared_molecules()
        if sm is None or sm.graph.num_nodes > 0.6 * ins.shape[0]:
            return None
        return sm

    def forward(self, ins, batch_data, model, out=None):
        """``out`` (optional) lets the caller place the result in a column slice of
        a wider buffer (multi-scale readout / GMN f_node input)."""
        ops.require_device(ins, "ins")
        if ag.training_active(self):
            return self._forward_train(ins, batch_data, model, out)
        csr = self._graph(batch_data).csr()
        act = act_name(self.act)
        slopes = prelu_slopes(self._cache, "prelu", self.act, self.out_dim)
        scale, shift = bn_affine(self._cache, "bn", self.bn)
        with torch.no_grad():
            plan = pool = None
            if self.type == 'gin' and not self.higher_level:
                if model.acts is not None and len(model.acts) == 1:      # first layer of a forward
                    model._readout_plan = _build_readout_plan(self, model, batch_data) if out is None else None
                plan = getattr(model, "_readout_plan", None)
                if plan is not None and (plan.batch is not batch_data or out is not None):
                    plan = model._readout_plan = None
                pool = plan.pool_for(self) if plan is not None else None
            fused_pool = False
            wide = None
            want_sp = False
            shared = shared_sp = None
            if self.type == 'gin' and out is None and not self.higher_level and not self.normalize \
                    and 'gmn' in FLAGS.model:
                # the GMNPropagator that follows reads this layer's output as the first D columns of its [n, 4D]
                # f_node input (model/layers_gmn.py:44-46): produce it there instead of copying it later
                from .layers_gmn import GMNPropagator
                if GMNPropagator._use_sp(ins.shape[0], self.out_dim):
                    # sp pipeline: the propagator's GEMMs read this layer's output as fp16 piece planes, written by
                    # the layer kernel's store loop next to the fp32 rows (which the attention kernel reads)
                    want_sp = True
                    shared = self._shared_molecules(ins, batch_data, model, pool)
                else:
                    wide = torch.empty((ins.shape[0], 4 * self.out_dim), dtype=torch.float32, device=ins.device)
                    out = wide[:, :self.out_dim]
            if self.type == 'gin':
                lin1, lin2 = self.conv.nn[0], self.conv.nn[2]
                w1 = self._cache.get("w1", lambda w: w.t().contiguous(), lin1.weight)
                w2 = self._cache.get("w2", lambda w: w.t().contiguous(), lin2.weight)
                eps = self._cache.get("eps", lambda e: float(e[0]), self.conv.eps)
                transform = getattr(FLAGS, "feature_transform", "bf16x6")
                g = self._graph(batch_data)
                if shared is not None:
                    # first layer of a pair batch: the rows of a molecule are the same in every pair it occurs in
                    # (this layer sees no cross-graph edge), so the layer runs on the distinct molecules
                    g = shared.graph
                    ins, csr = g.x, g.csr()
                codes = getattr(g, "codes", None) if ins is getattr(g, "x", None) else None
                if transform == "f16x3" and getattr(FLAGS, "categorical_input", True) and \
                        ops.gin_layer_h3_codes_supported(csr, codes, getattr(g, "code_offsets", None) or [0], self.out_dim) \
                        and g.code_offsets[-1] == ins.shape[1]:
                    # first lower layer on categorical features: 8 bytes of value indices per atom instead of the
                    # one-hot row (utils/data/representation_node_feat.py:86-97), expanded inside the kernel
                    w1p = self._cache.get("w1h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin1.weight)
                    w2p = self._cache.get("w2h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin2.weight)
                    x = ops.gin_layer_h3_codes(csr, codes, g.code_offsets, w1p, lin1.bias, w2p, lin2.bias,
                                               self.out_dim, eps=eps, act=act, act_param=slopes, bn_scale=scale,
                                               bn_shift=shift, out=out, pool=pool)
                    fused_pool = True
                elif transform == "f16x3" and out is None and wide is None and getattr(FLAGS, "gin_sp", True) and \
                        ops.gin_sp_preferred(csr, ins, self.out_dim, force=getattr(FLAGS, "gin_sp", True) == "force"):
                    # layers wider than the single-launch f16x3 kernel covers (d = 512), on large batches: aggregation
                    # straight into the sp format + the two Linears as pre-split GEMMs; the readout rides on the second
                    # GEMM's epilogue, and so do the sp planes a GMN propagator reads
                    n_rows, N = ins.shape[0], self.out_dim
                    agg_sp = ops.spmm_sum_sp(csr, ins, self_coef=1.0 + eps, skip_self=True)
                    h_sp = ops.SpRows(n_rows, N, ins.device)
                    ops.linear_sp(agg_sp, w1, lin1.bias, act=act, act_param=slopes, out_sp=h_sp)
                    f_sp = ops.SpRows(n_rows, 4 * N, ins.device) if want_sp else None      # (any whole-block width)
                    sp_x = f_sp.cols(0, N) if f_sp is not None else None
                    pool_here = pool if (pool is not None and sp_x is None and act in ("relu", "identity")) else None
                    x, _ = ops.linear_sp(h_sp, w2, lin2.bias, act=act, act_param=slopes, bn_scale=scale, bn_shift=shift,
                                         out_sp=sp_x, f32=True, pool=pool_here)
                    if sp_x is not None and shared is None:
                        x._blgnn_sp = f_sp
                    if shared is not None:
                        shared_sp = f_sp
                    fused_pool = pool_here is not None
                elif transform == "f16x3" and ops.gin_layer_h3_supported(csr, ins, ins.shape[1], self.out_dim):
                    # dense transforms on the fp16 matrix cores at fp32 accuracy (scaled two-piece split)
                    w1p = self._cache.get("w1h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin1.weight)
                    w2p = self._cache.get("w2h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin2.weight)
                    # (GMN models: the propagator's message GEMM normalises this output row by row; its row
                    # magnitudes come out of this kernel's store loop instead of a pass over the result)
                    amax = torch.empty(ins.shape[0], dtype=torch.float32, device=ins.device) if wide is not None else None
                    f_sp = ops.SpRows(ins.shape[0], 4 * self.out_dim, ins.device) \
                        if (want_sp and self.out_dim == ops.SpRows.BLOCK) else None
                    sp_x = f_sp.cols(0, self.out_dim) if f_sp is not None else None
                    x = ops.gin_layer_h3(csr, ins, w1p, lin1.bias, w2p, lin2.bias, self.out_dim, eps=eps,
                                         act=act, act_param=slopes, bn_scale=scale, bn_shift=shift, out=out, pool=pool,
                                         amax_out=amax, out_sp=sp_x)
                    if amax is not None:
                        x._blgnn_amax = amax
                    if sp_x is not None and shared is None:
                        x._blgnn_sp = f_sp
                    fused_pool = True
                    if shared is not None:
                        shared_sp = f_sp if sp_x is not None else None
                elif transform in ("bf16x6", "f16x3") and \
                        ops.gin_layer_x6_supported(csr, ins, ins.shape[1], self.out_dim):
                    # dense transforms on the bf16 matrix cores at fp32 accuracy (three-piece split)
                    w1p = self._cache.get("w1p", lambda w: ops.pack_weight_x6(w.t().contiguous()), lin1.weight)
                    w2p = self._cache.get("w2p", lambda w: ops.pack_weight_x6(w.t().contiguous()), lin2.weight)
                    x = ops.gin_layer_x6(csr, ins, w1p, lin1.bias, w2p, lin2.bias, self.out_dim, eps=eps,
                                         act=act, act_param=slopes, bn_scale=scale, bn_shift=shift, out=out)
                elif ops.gin_layer_supported(csr, ins, ins.shape[1], self.out_dim):
                    x = ops.gin_layer(csr, ins, w1, lin1.bias, w2, lin2.bias, eps=eps, act=act,
                                      act_param=slopes, bn_scale=scale, bn_shift=shift, out=out)
                else:
                    agg = ops.spmm_sum(csr, ins, self_coef=1.0 + eps, skip_self=True)
                    h = ops.linear(agg, w1, lin1.bias, act=act, act_param=slopes)
                    x = ops.linear(h, w2, lin2.bias, act=act, act_param=slopes,
                                   bn_scale=scale, bn_shift=shift, out=out)
            elif self.type == 'gat':
                # ``_out_rows`` (set by BiLevelEngine around the LAST interaction-level layer): only the rows the pair
                # scorer reads are computed
                x = ops.gat_conv(csr, ins, self.conv.weight, self.conv.att, self.conv.bias,
                                 group=getattr(FLAGS, "gat_softmax_group", "src"),
                                 slope=self.conv.negative_slope, act=act, act_param=slopes,
                                 bn_scale=scale, bn_shift=shift, out=out, rows=getattr(self, "_out_rows", None))
            else:
                # layers.py:48-50 reads edge_attr of the LOWER merge even for upper layers;
                # it is None in every shipped config, and only GCN would consume it.
                edge_weight = getattr(batch_data.merge_data['merge'], "edge_attr", None)
                x = ops.gcn_conv(csr, ins, self.conv.weight, self.conv.bias, edge_weight,
                                 act=act, act_param=slopes, bn_scale=scale, bn_shift=shift, out=out)
            if pool is not None:
                if not fused_pool:          # this layer's kernel has no readout epilogue: one segment-reduce launch
                    ops.segment_reduce(plan.seg, x, plan.kind, out=pool[2])
                plan.done.add(id(self))
            if wide is not None:
                x._blgnn_wide = wide
            if shared is not None:
                # replicate: the attention (and model.acts) read one row per merged atom; the propagator that follows
                # keeps working on the distinct molecules until its f_node input (layers_gmn.py: _forward_sp)
                x_u = x
                x = ops.expand_rows(x_u, shared.row_map)
                x._blgnn_shared = (shared, x_u, shared_sp)
            model.store_layer_output(self, x)
            if self.normalize:
                x = ops.l2_normalize_rows(x)
        return x

    def _forward_train(self, ins, batch_data, model, out=None):
        """Train mode (src/train.py:177-182 runs backward through this layer): the same arithmetic as
        ``forward`` unfused -- GIN: aggregation, Linear+act, Linear+act; GAT: transform, softmax-weighted gather + act;
        then train-mode BatchNorm -- as autograd nodes whose both directions are libblgnn kernels (blgnn/autograd.py)."""
        if out is not None:
            raise NotImplementedError("blgnn: train mode does not write into a caller-provided slice")
        csr = self._graph(batch_data).csr()
        act = act_name(self.act)
        # a PReLU trains its slopes: the layers run with the identity and the activation is its own node behind them
        # (it keeps the pre-activation); in the GIN layer the reference puts the SAME module between the two Linears
        # and behind the conv (model/layers.py:26-31,55), so both uses feed one slope gradient
        trained = self.act.weight if act == "prelu" else None
        if trained is not None:
            act = "identity"
        after = (lambda t: ag.prelu(t, trained)) if trained is not None else (lambda t: t)
        if self.type == 'gin':
            lin1, lin2 = self.conv.nn[0], self.conv.nn[2]
            # (eps is a buffer, not trained: read once per version, no device -> host copy per step)
            agg = ag.spmm_sum(ins, csr, 1.0 + self._cache.get("eps", lambda e: float(e[0]), self.conv.eps))
            h = after(ag.linear(agg, lin1.weight, lin1.bias, act))
            x = ag.linear(h, lin2.weight, lin2.bias, act)           # conv, then the layer's act (layers.py:55)
        elif self.type == 'gat':
            x = ag.gat_conv(ins, self.conv.weight, self.conv.att, self.conv.bias, csr,
                            group=getattr(FLAGS, "gat_softmax_group", "src"), slope=self.conv.negative_slope, act=act)
        else:
            # (model/layers.py:48-52: the LOWER merge's edge_attr is the edge weight, for upper layers too; data, no gradient)
            x = ag.gcn_conv(ins, self.conv.weight, self.conv.bias, csr, act=act,
                            edge_weight=getattr(batch_data.merge_data['merge'], "edge_attr", None))
        x = after(x)
        if self.bn:
            x = ag.batch_norm_train(x, self.bn)
        model.store_layer_output(self, x)
        if self.normalize:
            x = ag.l2_normalize(x)
        return x

    # ---- GAT split at its row-wise / graph-wise seam (multi-GPU: blgnn/engine.py) -------------------------
    def gat_transform_rows(self, x, h=None):
        """Row-wise half of a GAT layer on any subset of rows: -> (h, scores [2, rows])."""
        assert self.type == 'gat'
        with torch.no_grad():
            return ops.gat_transform(x, self.conv.weight, self.conv.att, h=h)

    def gat_forward_from_transform(self, h, sc, batch_data, model, out=None, rows=None):
        """Graph half of the layer on the full transformed table; same result as ``forward`` on the rows
        the table was transformed from.  ``rows`` (int64, device): only these target rows are computed (a rank's own
        rows of a target-row-sharded interaction level, or the rows the scored pairs read)."""
        assert self.type == 'gat'
        csr = self._graph(batch_data).csr()
        slopes = prelu_slopes(self._cache, "prelu", self.act, self.out_dim)
        scale, shift = bn_affine(self._cache, "bn", self.bn)
        with torch.no_grad():
            x = ops.gat_from_transform(csr, h, sc, self.conv.bias, group=getattr(FLAGS, "gat_softmax_group", "src"),
                                       slope=self.conv.negative_slope, act=act_name(self.act), act_param=slopes,
                                       bn_scale=scale, bn_shift=shift, out=out, rows=rows)
            model.store_layer_output(self, x)
            if self.normalize:
                x = ops.l2_normalize_rows(x)
        return x


class Loss(nn.Module):
    """model/layers.py:66-89.  After the hot path; plain torch."""

    def __init__(self, type):
        super(Loss, self).__init__()
        self.type = type
        if type == 'BCE':
            self.loss = nn.BCELoss()
        elif type == 'BCEWithLogits':
            self.loss = nn.BCEWithLogitsLoss()
        elif type == 'CE':
            self.loss = nn.CrossEntropyLoss()
        else:
            raise ValueError('Unknown loss layer type {}'.format(type))

    def forward(self, ins, batch_data, _):
        labels = batch_data.true_labels()
        if self.type != 'CE':
            return self.loss(ins.view(len(batch_data.batch_gids)), labels.float())
        return self.loss(ins, labels.long())
