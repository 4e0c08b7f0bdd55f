"""Training step of the hot path: autograd nodes whose forward AND backward are libblgnn kernels.

The reference trains with ``loss.backward(); optimizer.step()`` (src/train.py:177-182) on the same layer
stack the forward path runs.  Here ``torch.autograd`` only records the graph and routes gradients (plumbing);
every arithmetic step of both directions is a kernel behind the C ABI:

  Linear (+act)      dX = g @ W            blgnn_linear on nn.Linear's own [out, in] weight
                     dW = g^T @ X          blgnn_grad_weight (row chunks on exact-f32 MFMA, fixed-order sum)
                     db = sum_rows g       blgnn_col_reduce;   g = dY * act'(Y): blgnn_act_backward
  GIN aggregation    same CSR, same op     blgnn_spmm_sum (molecular graphs are symmetric)
  BatchNorm1d train  statistics / dgamma / dbeta: blgnn_col_reduce; blgnn_bn_apply / blgnn_bn_backward
  readout            blgnn_segment_broadcast per pooled activation
  pair scorer        blgnn_l2_normalize_backward, blgnn_pair_scatter_add

  GATConv            blgnn_gat_backward_scores / _features (+ blgnn_linear for datt, dW, dx)

Scope: GIN, GAT and GCN NodeEmbedding layers on either level, the avg_pool / sum readouts (multi-scale
or not, with the pooled rows entering the interaction graph's init_x), the mlp_concat pair scorer, BCE -- i.e. the
LL-GNN ("lower_level_gnn") and BI-GNN ("lower_level_gnn_higher_level") models end to end -- with relu / identity /
sigmoid / tanh activations; GCN layers (without edge weights) and the multi-edge-type GAT interaction level of DrugCombo
too, and the GMN propagator (message, cross-graph attention, f_node MLP with train-mode BatchNorm).
"""

import torch

from . import ops

__all__ = ["linear", "prelu", "gru_gates", "gat_conv", "gcn_conv", "multi_edge_gat", "multi_edge_gcn", "hadamard", "edge_relu_sum", "gmn_match", "spmm_sum", "batch_norm_train", "segment_reduce_multi", "l2_normalize", "pair_gather",
           "training_active", "set_batch_norm_sync"]


def training_active(module):
    """A module takes the autograd path when it is in train mode and gradients are being recorded."""
    return module.training and torch.is_grad_enabled()


def _gemm_transform(M):
    """The products of ``_Linear`` (y = x W^T and dx = g W): on row counts where the 16-bit matrix-core GEMMs run (the pair
    batches of the GMN configs: 479 k atoms) the arithmetic the forward path uses -- FLAGS.feature_transform, three fp16 /
    six bf16 MFMA products per fp32 product, fp32-equivalent --, exact-f32 MFMA below (the interaction level, small
    batches).  FLAGS.train_exact_gemm = True keeps exact-f32 MFMA everywhere."""
    from .config import FLAGS
    if getattr(FLAGS, "train_exact_gemm", False) or M < 32768:
        return "f32"
    return None


def _check_act(act):
    if act == "prelu":
        raise ValueError("blgnn: a trained PReLU is its own autograd node (autograd.prelu) behind an identity-activation layer")


class _PRelu(torch.autograd.Function):
    """y = PReLU(z), slope [1] or [N] (create_act('prelu', out_dim): model/layers_util.py:63-64, the activation of a
    NodeEmbedding layer -- model/layers.py:24,26,34,55; inside the GIN MLP the SAME module, so its slope gradient is the
    sum over both uses, which autograd accumulates).  The pre-activation is kept: the derivative tests ITS sign, so the
    slopes may take any sign while training."""

    @staticmethod
    def forward(ctx, z, slope):
        z, slope = z.detach(), slope.detach()
        per_col = slope.expand(z.shape[1]).contiguous()
        y = ops.act_forward(z, "prelu", per_col)
        ctx.single = slope.numel() == 1 and z.shape[1] != 1
        ctx.save_for_backward(z, per_col)
        return y

    @staticmethod
    def backward(ctx, dy):
        z, per_col = ctx.saved_tensors
        dy = dy.contiguous()
        dz = ops.act_backward(dy, z, "prelu", per_col) if ctx.needs_input_grad[0] else None
        ds = None
        if ctx.needs_input_grad[1]:
            ds = ops.col_reduce(dy, "dot_neg", b=z)                               # sum_m dy * min(z, 0) per channel
            if ctx.single:
                ds = ops.col_reduce(ds.view(-1, 1), "sum")                        # one shared slope: nn.PReLU()
        return dz, ds


def prelu(z, slope):
    return _PRelu.apply(z, slope)


class _GruGates(torch.autograd.Function):
    """h' = GRUCell's gate arithmetic on gi = in W_ih^T + b_ih, gh = h W_hh^T + b_hh and the state h (torch.nn.GRUCell;
    GMNPropagator(f_node='GRU'): model/layers_gmn.py:27-30,48-50).  The two products are ``linear`` nodes around it."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        gi, gh, h = gi.detach(), gh.detach(), h.detach()
        ctx.save_for_backward(gi, gh, h)
        return ops.gru_gates(gi, gh, h)

    @staticmethod
    def backward(ctx, g):
        gi, gh, h = ctx.saved_tensors
        dgi, dgh, dh = ops.gru_gates_backward(gi, gh, h, g.contiguous())
        return dgi, dgh, dh


def gru_gates(gi, gh, h):
    return _GruGates.apply(gi, gh, h)


class _Linear(torch.autograd.Function):
    """y = act(x @ weight^T + bias), weight in nn.Linear's [out, in] layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        _check_act(act)
        w_kn = ops.transpose(weight.detach().contiguous())
        y = ops.linear(x.detach(), w_kn, None if bias is None else bias.detach(), act=act,
                       transform=_gemm_transform(x.shape[0]))
        ctx.act = act
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x.detach(), weight.detach(), y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        g = dy if ctx.act == "identity" else ops.act_backward(dy, y, ctx.act)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(g, weight.contiguous(), transform=_gemm_transform(g.shape[0]))   # [M, out] @ [out, in]
        if ctx.needs_input_grad[1]:
            dw = ops.grad_weight(g, x)                                            # sum_m g[m, out] x[m, in]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.col_reduce(g, "sum")
        return dx, dw, db, None


def linear(x, weight, bias=None, act="identity"):
    return _Linear.apply(x, weight, bias, act)


class _GatConv(torch.autograd.Function):
    """y = act(GATConv(x) ), heads = 1 (model/layers.py:33,54): weight [in, out], att [1, 1, 2 out], bias [out]."""

    @staticmethod
    def forward(ctx, x, weight, att, bias, csr, group, slope, act):
        _check_act(act)
        x, weight, att = x.detach(), weight.detach(), att.detach()
        csr.require_symmetric("GATConv backward")
        y, h, sc, stats = ops.gat_forward_saved(csr, x, weight, att, bias.detach(), group, slope, act)
        ctx.csr, ctx.group, ctx.slope, ctx.act = csr, group, slope, act
        ctx.save_for_backward(x, weight, att, h, sc, stats, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, att, h, sc, stats, y = ctx.saved_tensors
        dy = dy.contiguous()
        g = dy if ctx.act == "identity" else ops.act_backward(dy, y, ctx.act)
        dh, ds = ops.gat_backward(ctx.csr, h, sc, stats, g, att, ctx.group, ctx.slope)
        dx = dw = datt = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear(dh, ops.transpose(weight.contiguous()), transform="f32")        # [n, out] @ [out, in]
        if ctx.needs_input_grad[1]:
            dw = ops.grad_weight(x, dh)                                                     # [in, out]
        if ctx.needs_input_grad[2]:
            datt = ops.linear(ds, h, transform="f32").reshape(att.shape)                    # [2, n] @ [n, out]
        if ctx.needs_input_grad[3]:
            db = ops.col_reduce(g, "sum")
        return dx, dw, datt, db, None, None, None, None


def gat_conv(x, weight, att, bias, csr, group="src", slope=0.2, act="identity"):
    return _GatConv.apply(x, weight, att, bias, csr, group, slope, act)


def _symmetric_edge_weights(csr, w):
    """True when the per-edge weights ``w`` (CSR order) satisfy w(i, j) == w(j, i) on the symmetric graph ``csr`` (index
    arithmetic on the device, one host sync; called once per training forward that brings edge weights)."""
    counts = (csr.rowptr[1:] - csr.rowptr[:-1]).long()
    rows = torch.repeat_interleave(torch.arange(csr.n, device=csr.col.device), counts)
    cols = csr.col.long()
    fwd = torch.argsort(rows * csr.n + cols)
    bwd = torch.argsort(cols * csr.n + rows)
    return bool(torch.equal(w[fwd], w[bwd]))


class _GcnConv(torch.autograd.Function):
    """y = act(A_hat (x W) + b), GCNConv (model/layers.py:23,48-52); A_hat = D^-1/2 (A_w + I) D^-1/2 is symmetric -- without
    edge weights, or with weights that are the same in both directions of an edge --, so the gradient of the propagation
    is the propagation itself.  The weights are data (no gradient); asymmetric ones are refused."""

    @staticmethod
    def forward(ctx, x, weight, bias, csr, act, edge_weight=None):
        _check_act(act)
        x, weight = x.detach(), weight.detach()
        csr.require_symmetric("GCNConv backward")
        if edge_weight is not None:
            edge_weight = edge_weight.detach().reshape(-1).contiguous()
            if edge_weight.numel() != csr.nnz:
                raise ValueError("blgnn: GCNConv expects one weight per edge (%d), got %d" % (csr.nnz, edge_weight.numel()))
            if not _symmetric_edge_weights(csr, edge_weight):
                raise ValueError("blgnn: GCNConv backward needs w(i, j) == w(j, i): the backward pass reuses the forward "
                                 "operator as its transpose")
        y = ops.gcn_conv(csr, x, weight, bias.detach(), edge_weight, act=act)
        ctx.csr, ctx.act, ctx.edge_weight = csr, act, edge_weight
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = dy.contiguous()
        g = dy if ctx.act == "identity" else ops.act_backward(dy, y, ctx.act)
        dh = ops.gcn_propagate(ctx.csr, g, ctx.edge_weight)
        dx = ops.linear(dh, ops.transpose(weight.contiguous()), transform="f32") if ctx.needs_input_grad[0] else None
        dw = ops.grad_weight(x, dh) if ctx.needs_input_grad[1] else None
        db = ops.col_reduce(g, "sum") if ctx.needs_input_grad[2] else None
        return dx, dw, db, None, None, None


def gcn_conv(x, weight, bias, csr, act="identity", edge_weight=None):
    return _GcnConv.apply(x, weight, bias, csr, act, edge_weight)


class _MultiEdgeGat(torch.autograd.Function):
    """y = act( sum_k GATConv_k(x, edges_k) ): NodeModelAggrByEdge('gat') of the DrugCombo interaction level
    (model/layers_meta.py:70,74-79).  params = (weight_0, att_0, bias_0, weight_1, ...)."""

    @staticmethod
    def forward(ctx, x, csrs, group, slope, act, *params):
        _check_act(act)
        x = x.detach()
        K = len(csrs)
        for c in csrs:
            c.require_symmetric("multi-edge GAT backward")
        saved, y = [], None
        for k in range(K):
            w, a, b = (p.detach() for p in params[3 * k:3 * k + 3])
            # every edge type accumulates into the same buffer; the activation rides on the last pass
            y, h, sc, stats = ops.gat_forward_saved(csrs[k], x, w, a, b, group, slope, act if k == K - 1 else "identity",
                                                    out=y, accumulate=k > 0)
            saved += [w, a, h, sc, stats]
        ctx.csrs, ctx.group, ctx.slope, ctx.act, ctx.K = csrs, group, slope, act, K
        ctx.save_for_backward(x, y, *saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors[:2]
        saved = ctx.saved_tensors[2:]
        dy = dy.contiguous()
        g = dy if ctx.act == "identity" else ops.act_backward(dy, y, ctx.act)
        db = ops.col_reduce(g, "sum")                                     # every edge type's bias sees the same g
        grads, dhs, wts = [], [], []
        for k in range(ctx.K):
            w, a, h, sc, stats = saved[5 * k:5 * k + 5]
            dh, ds = ops.gat_backward(ctx.csrs[k], h, sc, stats, g, a, ctx.group, ctx.slope)
            grads += [ops.grad_weight(x, dh), ops.linear(ds, h, transform="f32").reshape(a.shape), db]
            dhs.append(dh)
            wts.append(ops.transpose(w.contiguous()))
        dx = None
        if ctx.needs_input_grad[0]:
            # sum_k dh_k W_k^T as ONE product over the concatenated inner dimension
            dx = ops.linear(torch.cat(dhs, dim=1), torch.cat(wts, dim=0), transform="f32")
        return (dx, None, None, None, None) + tuple(grads)


def multi_edge_gat(x, convs, csrs, group="src", slope=0.2, act="identity"):
    params = []
    for c in convs:
        params += [c.weight, c.att, c.bias]
    return _MultiEdgeGat.apply(x, list(csrs), group, slope, act, *params)


class _MultiEdgeGcn(torch.autograd.Function):
    """y = act( sum_k GCNConv_k(x, edges_k) ): NodeModelAggrByEdge('gcn') (model/layers_meta.py:68,74-79).
    params = (weight_0, bias_0, weight_1, ...), weights in GCNConv's [in, out] layout.  Every A_hat_k is symmetric, so
    each edge type's propagation is its own gradient operator."""

    @staticmethod
    def forward(ctx, x, csrs, act, *params):
        _check_act(act)
        x = x.detach()
        K = len(csrs)
        for c in csrs:
            c.require_symmetric("multi-edge GCN backward")
        y = None
        for k in range(K):
            w, b = (p.detach() for p in params[2 * k:2 * k + 2])
            # every edge type accumulates into the same buffer; the activation rides on the last pass
            y = ops.gcn_conv(csrs[k], x, w, b, None, act=act if k == K - 1 else "identity", out=y, accumulate=k > 0)
        ctx.csrs, ctx.act, ctx.K = csrs, act, K
        ctx.save_for_backward(x, y, *[p.detach() for p in params[0::2]])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors[:2]
        ws = ctx.saved_tensors[2:]
        dy = dy.contiguous()
        g = dy if ctx.act == "identity" else ops.act_backward(dy, y, ctx.act)
        db = ops.col_reduce(g, "sum")                                     # every edge type's bias sees the same g
        grads, dhs, wts = [], [], []
        for k in range(ctx.K):
            dh = ops.gcn_propagate(ctx.csrs[k], g)
            grads += [ops.grad_weight(x, dh), db]
            dhs.append(dh)
            wts.append(ops.transpose(ws[k].contiguous()))
        dx = None
        if ctx.needs_input_grad[0]:
            # sum_k dh_k W_k^T as ONE product over the concatenated inner dimension
            dx = ops.linear(torch.cat(dhs, dim=1), torch.cat(wts, dim=0), transform="f32")
        return (dx, None, None) + tuple(grads)


def multi_edge_gcn(x, convs, csrs, act="identity"):
    params = []
    for c in convs:
        params += [c.weight, c.bias]
    return _MultiEdgeGcn.apply(x, list(csrs), act, *params)


class _Hadamard(torch.autograd.Function):
    """y = a * b, element by element: the gate of GMNAggregatorPairs (model/layers_aggregation.py:91-93) and the
    'dot_product' pair scorer (model/layers_link_pred.py:65-66)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.detach(), b.detach()
        ctx.save_for_backward(a, b)
        return ops.hadamard(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return (ops.hadamard(dy, b) if ctx.needs_input_grad[0] else None,
                ops.hadamard(dy, a) if ctx.needs_input_grad[1] else None)


def hadamard(a, b):
    return _Hadamard.apply(a, b)


class _SpmmSum(torch.autograd.Function):
    """agg_i = self_coef * x_i + sum_{j in N(i)} x_j on a SYMMETRIC CSR: the gradient is the same operator."""

    @staticmethod
    def forward(ctx, x, csr, self_coef):
        csr.require_symmetric("GIN aggregation backward")
        ctx.csr, ctx.self_coef = csr, self_coef
        return ops.spmm_sum(csr, x.detach(), self_coef=self_coef, skip_self=True)

    @staticmethod
    def backward(ctx, dy):
        return ops.spmm_sum(ctx.csr, dy.contiguous(), self_coef=ctx.self_coef, skip_self=True), None, None


def spmm_sum(x, csr, self_coef):
    return _SpmmSum.apply(x, csr, self_coef)


_SYNC = {"reduce": None}


def set_batch_norm_sync(reduce_fn):
    """Data-parallel training (one process per GPU, each rank scoring its own pairs): train-mode BatchNorm must see the
    statistics of the WHOLE batch.  ``reduce_fn(t)`` sums the float tensor ``t`` over the ranks in place and returns it
    (``lambda t: (dist.all_reduce(t, group=g), t)[1]`` over RCCL); ``None`` switches synchronisation off.  What crosses
    the wire per BatchNorm layer and direction is 2 N + 1 floats."""
    _SYNC["reduce"] = reduce_fn


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, stats_out, act="identity", running=None):
        _check_act(act)
        x = x.detach()
        M, N = x.shape
        red = _SYNC["reduce"]
        rstd = None
        if red is None:
            count = float(M)
            mean = ops.col_reduce(x, "sum", scale=1.0 / M)
            var = ops.col_reduce(x, "sqdev", mean=mean, scale=1.0 / M)            # biased, two-pass
            if running is not None:
                # rstd and torch's running-statistics update (momentum form, unbiased variance) in one launch
                rm, rv, mom = running
                rstd = ops.bn_finalize(mean, var, eps, mom, count / max(count - 1.0, 1.0), rm, rv)
                # the kernel wrote rm / rv through raw pointers: the eval-BN fold is cached on their versions
                torch.autograd.graph.increment_version((rm, rv))
        else:
            # two-pass over the union of the ranks' rows: global sum -> mean, global centred squares -> variance
            buf = torch.empty(N + 1, dtype=torch.float32, device=x.device)
            ops.col_reduce(x, "sum", out=buf[:N])
            buf[N] = float(M)
            buf = red(buf)
            count = float(buf[N])
            mean = (buf[:N] / count).contiguous()
            var = red(ops.col_reduce(x, "sqdev", mean=mean)) / count
        fused = rstd is not None
        if not fused:
            rstd = torch.rsqrt(var + eps)                                         # [N] vector glue
        y = ops.bn_apply(x, mean, rstd, None if gamma is None else gamma.detach(),
                         None if beta is None else beta.detach(), act=act)
        stats_out.extend((mean, var, count, fused))                               # fused: running statistics already updated
        ctx.save_for_backward(x, mean, rstd, gamma.detach() if gamma is not None else None,
                              y if act != "identity" else None)
        ctx.has_affine = gamma is not None
        ctx.count = count
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd, gamma, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act != "identity":
            dy = ops.act_backward(dy, y, ctx.act)
        N = x.shape[1]
        both = torch.empty((2, N), dtype=torch.float32, device=x.device)
        dbeta = ops.col_reduce(dy, "sum", out=both[0])
        dgamma = ops.col_reduce(dy, "dot_norm", b=x, mean=mean, rstd=rstd, out=both[1])     # sum dy * xhat
        dx = None
        if ctx.needs_input_grad[0]:
            red = _SYNC["reduce"]
            tot = both if red is None else red(both.clone())                     # dx needs the sums over ALL rows
            dx = ops.bn_backward(dy, x, mean, rstd, gamma, tot[1], tot[0], count=ctx.count)
        # the parameter gradients stay local sums (a data-parallel wrapper reduces parameter gradients itself)
        return dx, (dgamma if ctx.has_affine else None), (dbeta if ctx.has_affine else None), None, None, None, None


def batch_norm_train(x, bn, act="identity"):
    """torch.nn.BatchNorm1d in train mode (model/layers.py:39-40,56-57), including the running-statistics
    update (momentum form, unbiased variance) torch performs as a side effect.  ``act``: the activation that
    follows the BatchNorm inside an MLP (model/layers_util.py:51-54), fused into the same kernels."""
    stats = []
    track = bn.track_running_stats and bn.running_mean is not None
    running = None
    if track and bn.momentum is not None:
        running = (bn.running_mean, bn.running_var, float(bn.momentum))           # updated inside the forward's launch
    y = _BatchNormTrain.apply(x, bn.weight, bn.bias, bn.eps, stats, act, running)
    if track:
        mean, var, M, updated = stats
        with torch.no_grad():
            bn.num_batches_tracked += 1
            if not updated:                                                       # (rank-synchronised statistics, or a
                mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)   # cumulative average)
                bn.running_mean.mul_(1.0 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1.0 - mom).add_(var * (M / max(M - 1.0, 1.0)), alpha=mom)
    return y


class _EdgeReluSum(torch.autograd.Function):
    """S_i = sum_{j in N(i)} relu(P_i + Q_j): the restructured GMN message (model/layers_gmn.py:38-42)."""

    @staticmethod
    def forward(ctx, p, q, csr):
        p, q = p.detach().contiguous(), q.detach().contiguous()
        csr.require_symmetric("GMN message backward")
        ctx.csr = csr
        ctx.save_for_backward(p, q)
        return ops.edge_relu_sum(csr, p, q)

    @staticmethod
    def backward(ctx, ds):
        p, q = ctx.saved_tensors
        dp, dq = ops.edge_relu_sum_backward(ctx.csr, p, q, ds.contiguous())
        return dp, dq, None


def edge_relu_sum(p, q, csr):
    return _EdgeReluSum.apply(p, q, csr)


class _GmnMatch(torch.autograd.Function):
    """u = x - cross-graph attention of the partner graph (model/layers_gmn.py:55-84), pairs = consecutive graphs."""

    @staticmethod
    def forward(ctx, x, seg):
        x = x.detach().contiguous()
        ctx.seg = seg
        ctx.save_for_backward(x)
        return ops.gmn_match(seg, x)

    @staticmethod
    def backward(ctx, du):
        (x,) = ctx.saved_tensors
        return ops.gmn_match_backward(ctx.seg, x, du.contiguous()), None


def gmn_match(x, seg):
    return _GmnMatch.apply(x, seg)


class _SegmentReduceMulti(torch.autograd.Function):
    """torch.cat([scatter_mean(x, batch) for x in xs], dim=1) and its gradient."""

    @staticmethod
    def forward(ctx, seg, kind, row_map, *xs):
        ctx.seg, ctx.kind, ctx.row_map = seg, kind, row_map
        ctx.shapes = [tuple(x.shape) for x in xs]
        xs = [x.detach() for x in xs]
        if len(xs) <= 8:
            return ops.segment_reduce_multi(seg, xs, kind, row_map=row_map)
        G = seg.numel() - 1
        out = torch.empty((G, sum(s[1] for s in ctx.shapes)), dtype=torch.float32, device=xs[0].device)
        c = 0
        for x in xs:
            ops.segment_reduce(seg, x, kind, out=out[:, c:c + x.shape[1]], row_map=row_map)
            c += x.shape[1]
        return out

    @staticmethod
    def backward(ctx, dp):
        grads, c = [], 0
        if dp.stride(1) != 1:
            dp = dp.contiguous()
        for i, (n, d) in enumerate(ctx.shapes):
            if ctx.needs_input_grad[3 + i]:
                grads.append(ops.segment_broadcast(ctx.seg, dp[:, c:c + d], n, ctx.kind, row_map=ctx.row_map))
            else:
                grads.append(None)
            c += d
        return (None, None, None) + tuple(grads)


def segment_reduce_multi(seg, xs, kind="mean", row_map=None):
    return _SegmentReduceMulti.apply(seg, kind, row_map, *xs)


class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, eps):
        z = z.detach()
        ctx.eps = eps
        ctx.save_for_backward(z)
        return ops.l2_normalize_rows(z, eps)

    @staticmethod
    def backward(ctx, dzn):
        (z,) = ctx.saved_tensors
        return ops.l2_normalize_backward(z, dzn.contiguous(), ctx.eps), None


def l2_normalize(z, eps=1e-12):
    return _L2Normalize.apply(z, eps)


class _PairGather(torch.autograd.Function):
    """cat[b] = [zn[ids1[b]], zn[ids2[b]]]  (model/layers_link_pred.py:56-61); gradient = scatter-add."""

    @staticmethod
    def forward(ctx, zn, ids1, ids2):
        ctx.ids = (ids1, ids2)
        ctx.shape = tuple(zn.shape)
        zn = zn.detach()
        d = zn.shape[1]
        cat = torch.empty((ids1.numel(), 2 * d), dtype=torch.float32, device=zn.device)
        cat[:, :d] = zn.index_select(0, ids1)                  # row copies: data movement only
        cat[:, d:] = zn.index_select(0, ids2)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        dcat = dcat.contiguous()
        dz = torch.zeros(ctx.shape, dtype=torch.float32, device=dcat.device)
        d = ctx.shape[1]
        ops.pair_scatter_add(dcat, 0, ctx.ids[0], dz)
        ops.pair_scatter_add(dcat, d, ctx.ids[1], dz)
        return dz, None, None


def pair_gather(zn, ids1, ids2):
    return _PairGather.apply(zn, ids1, ids2)
