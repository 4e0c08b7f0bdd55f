"""TEST-ONLY stand-in for ``blgnn.ops`` on CPU tensors, built from the oracle.

Purpose: exercise the HOST logic of the drop-in modules (spec parsing, parameter
naming, layer wiring, column-slice outputs, BatchData / MoleculeStore, the
distributed sharding) in the GPU-less container and under ``gloo``.  It checks
plumbing only -- both sides of such a test use oracle arithmetic -- and is never
importable from the product package (tests/test_abi.py enforces that).  Kernel
parity is established by the ``-m gpu`` tests, which call libblgnn through the C ABI.
"""

import numpy as np
import torch

import oracle
from blgnn import ops as real_ops


class FakeCSR(object):
    def __init__(self, edge_index, n):
        self.key, self.nbr = edge_index[0], edge_index[1]
        self.n, self.nnz = n, edge_index.shape[1]
        self._deg = None

    @property
    def edge_index(self):          # messages flow nbr -> key
        return torch.stack([self.nbr, self.key])


def csr_from_edge_index(edge_index, num_nodes, symmetric=True):
    ei = edge_index if symmetric else torch.stack([edge_index[1], edge_index[0]])
    return FakeCSR(ei, num_nodes)


def segments_from_batch(batch, num_graphs):
    cnt = torch.bincount(batch, minlength=num_graphs)
    return torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(cnt, 0)]).to(torch.int32)


def _batch_of(seg):
    sizes = (seg[1:] - seg[:-1]).long()
    return torch.repeat_interleave(torch.arange(sizes.numel()), sizes), sizes.numel()


def _put(out, val):
    if out is None:
        return val
    out.copy_(val)
    return out


def _epi(v, act, act_param, bn_scale, bn_shift):
    v = oracle.apply_act(v, act, act_param)
    if bn_scale is not None:
        v = v * bn_scale + (bn_shift if bn_shift is not None else 0)
    return v


def spmm_sum(csr, x, self_coef=0.0, skip_self=False, w=None, out=None):
    ei = csr.edge_index
    keep = (ei[0] != ei[1]) if skip_self else torch.ones(ei.shape[1], dtype=torch.bool)
    msg = x[ei[0][keep]]
    if w is not None:
        msg = msg * w[keep].view(-1, 1)
    return _put(out, self_coef * x + oracle.scatter_add(msg, ei[1][keep], x.shape[0]))


def linear(x, w_kn, bias=None, act="identity", act_param=None, bn_scale=None, bn_shift=None,
           rowvec=None, rowvec_w=None, out=None, transform=None, row_amax=None, amax_out=None):
    if act not in real_ops.ACT_IDS:
        raise ValueError("Unknown activation function {}".format(act))
    v = x @ w_kn
    if bias is not None:
        v = v + bias
    if rowvec is not None:
        v = v + rowvec.view(-1, 1) * rowvec_w.view(1, -1)
    return _put(out, _epi(v, act, act_param, bn_scale, bn_shift))


def linear_writes_amax(x, w_kn, transform=None):
    return False


def gin_layer_h3_codes_supported(csr, codes, offsets, N):
    return False


def row_absmax(x, out=None):
    return x.abs().amax(dim=1)


def gin_layer_supported(csr, x, K, N):
    return N in (64, 128, 256) and K % 4 == 0 and K <= 256 and csr.nnz < 12 * max(csr.n, 1)


def gin_layer(csr, x, w1_kn, b1, w2_kn, b2, eps=0.0, act="identity", act_param=None,
              bn_scale=None, bn_shift=None, out=None):
    v = oracle.gin_conv(x, csr.edge_index, w1_kn.t(), b1, w2_kn.t(), b2, act, eps, act_param)
    return _put(out, _epi(v, act, act_param, bn_scale, bn_shift))


def gin_layer_h3_supported(csr, x, K, N):
    return False


def gin_layer_x6_supported(csr, x, K, N):
    return False                                   # the CPU stand-in has no packed-weight path


def gat_conv(csr, x, weight, att, bias, group="src", slope=0.2, act="identity", act_param=None,
             bn_scale=None, bn_shift=None, out=None, accumulate=False, rows=None):
    v = oracle.gat_conv(x, csr.edge_index, weight, att, bias, group, slope)
    if accumulate:
        v = v + out
    return _put(out, _epi(v, act, act_param, bn_scale, bn_shift))


def gcn_conv(csr, x, weight, bias, edge_weight=None, act="identity", act_param=None,
             bn_scale=None, bn_shift=None, out=None, accumulate=False):
    v = oracle.gcn_conv(x, csr.edge_index, weight, bias, edge_weight)
    if accumulate:
        v = v + out
    return _put(out, _epi(v, act, act_param, bn_scale, bn_shift))


def segment_reduce(seg, x, kind="mean", out=None, row_map=None):
    batch, G = _batch_of(seg)
    f = oracle.scatter_mean if kind == "mean" else oracle.scatter_add
    v = f(x, batch, G)
    if out is None:
        out = torch.empty((G, x.shape[1]))
    if row_map is not None:
        out[row_map.long()] = v
    else:
        out.copy_(v)
    return out


def segment_reduce_multi(seg, xs, kind="mean", row_map=None, out=None):
    return _put(out, torch.cat([segment_reduce(seg, x, kind) for x in xs], dim=1))


def gmn_match(seg, x, out=None, row_amax=None, max_nodes=0, classes=0, order=None):
    s = seg.tolist()
    return _put(out, oracle.gmn_match(x, [(s[i], s[i + 1]) for i in range(len(s) - 1)]))


def edge_relu_sum(csr, p, q, out=None, row_amax=None):
    return _put(out, oracle.scatter_add(torch.relu(p[csr.key] + q[csr.nbr]), csr.key, p.shape[0]))


def row_degree(csr):
    return torch.bincount(csr.key, minlength=csr.n).float()


def pair_gather_normalize(z, ids1, ids2, eps=1e-12):
    zn = oracle.l2_normalize(z, eps)
    return torch.cat([zn[ids1], zn[ids2]], dim=1)


def link_pred_head(z, ids1, ids2, weights_kn, biases, hidden_act="relu", final_act="identity", eps=1e-12):
    v = pair_gather_normalize(z, ids1, ids2, eps)
    for i, (w, b) in enumerate(zip(weights_kn, biases)):
        v = v @ w + (b if b is not None else 0)
        v = oracle.apply_act(v, hidden_act if i < len(weights_kn) - 1 else final_act)
    return v


def l2_normalize_rows(x, eps=1e-12, out=None):
    return _put(out, oracle.l2_normalize(x, eps))


def scatter_rows(src, row_map, dst):
    dst[row_map.long()] = src
    return dst


def require_device(t, name="tensor"):
    return None


def hadamard(a, b, out=None):
    return _put(out, a * b)


def col_reduce(a, mode="sum", b=None, mean=None, rstd=None, scale=1.0, out=None):
    if mode == "sum":
        v = a.sum(0)
    elif mode == "dot":
        v = (a * b).sum(0)
    elif mode == "sqdev":
        v = ((a - mean) ** 2).sum(0)
    else:
        v = (a * (b - mean) * rstd).sum(0)
    return _put(out, v * scale)


def bn_apply(x, mean, rstd, gamma=None, beta=None, out=None, act="identity"):
    y = (x - mean) * rstd
    if gamma is not None:
        y = y * gamma
    if beta is not None:
        y = y + beta
    return _put(out, oracle.apply_act(y, act, None))


def bn_finalize(mean, var, eps, momentum=0.0, unbias=1.0, running_mean=None, running_var=None):
    if running_mean is not None:
        running_mean.mul_(1.0 - momentum).add_(mean, alpha=momentum)
    if running_var is not None:
        running_var.mul_(1.0 - momentum).add_(var * unbias, alpha=momentum)
    return torch.rsqrt(var + eps)


def bn_backward(dy, x, mean, rstd, gamma, dgamma, dbeta, out=None, count=None):
    m = float(count if count is not None else x.shape[0])
    xh = (x - mean) * rstd
    g = gamma if gamma is not None else 1.0
    return _put(out, g * rstd * (dy - dbeta / m - xh * dgamma / m))


NAMES = ["csr_from_edge_index", "segments_from_batch", "spmm_sum", "linear", "linear_writes_amax", "row_absmax",
         "gin_layer_h3_codes_supported",
         "gin_layer",
         "gin_layer_supported", "gin_layer_x6_supported", "gat_conv", "gcn_conv",
         "segment_reduce", "segment_reduce_multi", "gmn_match", "edge_relu_sum", "row_degree", "pair_gather_normalize", "link_pred_head",
         "l2_normalize_rows", "scatter_rows", "require_device", "col_reduce", "bn_apply", "bn_finalize", "bn_backward", "hadamard"]


def install(monkeypatch):
    """Patch ``blgnn.ops`` in place for the duration of one test."""
    import sys
    mod = sys.modules[__name__]
    for n in NAMES:
        monkeypatch.setattr(real_ops, n, getattr(mod, n))
