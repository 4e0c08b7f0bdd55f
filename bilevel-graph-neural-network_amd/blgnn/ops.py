"""Tensor-level wrappers over the C ABI (include/blgnn.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every
arithmetic step is a libblgnn kernel.  All functions require CUDA(HIP) fp32
tensors and raise on anything else -- there is deliberately no CPU path.
"""

import weakref

import os

import torch

from . import _lib
from ._lib import call, ACT_IDS, GAT_GROUP, REDUCE_MEAN, REDUCE_SUM
from .config import FLAGS

__all__ = [
    "CSR", "csr_from_edge_index", "segments_from_batch", "spmm_sum", "linear",
    "gat_conv", "gat_transform", "gat_from_transform", "gcn_conv", "segment_reduce", "segment_reduce_multi", "gmn_match", "edge_relu_sum", "row_absmax", "linear_writes_amax", "linear_h3_preferred",
    "row_degree", "link_pred_head", "gat_aggregate", "pack_weight_x6", "pack_weight_h3", "pack_weight_h3g", "gin_layer_h3", "gin_layer_h3_supported", "gin_layer_h3_codes", "gin_layer_h3_codes_supported", "linear_x6_supported", "gin_layer_x6", "gin_layer_x6_supported", "gin_stack", "molecule_tiles", "gin_layer", "gin_layer_supported", "pair_gather_normalize", "l2_normalize_rows", "scatter_rows",
    "require_device",
]


def require_device(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            "blgnn: %s must live on an MI355X (cuda) device; the hot path has no CPU "
            "fallback (got %s)" % (name, getattr(t, "device", type(t))))


def _mat(t, name):
    """fp32 row-major [rows, cols] view with unit column stride; returns (ptr, ld)."""
    require_device(t, name)
    if t.dtype != torch.float32 or t.dim() != 2:
        raise ValueError("blgnn: %s must be a 2-D float32 tensor" % name)
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError("blgnn: %s must have unit column stride" % name)
    ld = t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
    if ld < t.shape[1]:
        raise ValueError("blgnn: %s rows overlap (stride %d < %d)" % (name, ld, t.shape[1]))
    return t.data_ptr(), ld


def _vec(t, name, dtype=torch.float32, optional=True):
    if t is None:
        if optional:
            return None
        raise ValueError("blgnn: %s is required" % name)
    require_device(t, name)
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError("blgnn: %s must be contiguous %s" % (name, dtype))
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _act(name):
    if name not in ACT_IDS:
        raise ValueError("Unknown activation function {}".format(name))
    return ACT_IDS[name]


class CSR(object):
    """int32 CSR on the device: row = aggregation target, col = sources."""

    __slots__ = ("rowptr", "col", "n", "nnz", "_deg", "_symmetric")

    def __init__(self, rowptr, col, n, nnz):
        self.rowptr, self.col, self.n, self.nnz = rowptr, col, int(n), int(nnz)
        self._deg = None
        self._symmetric = None

    def require_symmetric(self, what):
        """The backward kernels of the aggregation operators reuse the forward CSR as its own transpose
        (blgnn/autograd.py), which is only right for a symmetric edge set -- every graph of the reference path is
        (model/layers_util.py:133 to_undirected), but a caller-supplied edge_index need not be.  Checked once per
        CSR object on the device (index arithmetic, one host sync), cached; raises on an asymmetric graph instead
        of returning silently wrong gradients."""
        if self._symmetric is None:
            if self.nnz == 0:
                self._symmetric = True
            else:
                counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
                rows = torch.repeat_interleave(torch.arange(self.n, device=self.col.device), counts)
                cols = self.col.long()
                fwd = torch.sort(rows * self.n + cols).values
                bwd = torch.sort(cols * self.n + rows).values
                self._symmetric = bool(torch.equal(fwd, bwd))
        if not self._symmetric:
            raise ValueError("blgnn: %s needs a symmetric graph (edge (i, j) present iff (j, i) is): the backward "
                             "pass reuses the forward CSR as its transpose" % what)


def csr_from_edge_index(edge_index, num_nodes, symmetric=True):
    """edge_index: int64 [2, nnz] on the device (the reference layout).  With
    ``symmetric`` (every graph on this path: model/layers_util.py:133) the CSR is
    keyed by edge_index[0], which is already sorted -> no atomics, no row sort."""
    require_device(edge_index, "edge_index")
    if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
        raise ValueError("blgnn: edge_index must be int64 [2, nnz]")
    ei = edge_index.contiguous()
    nnz = ei.shape[1]
    key, nbr = (ei[0], ei[1]) if symmetric else (ei[1], ei[0])
    dev = ei.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
    work = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    call("blgnn_csr_from_coo", key.data_ptr(), nbr.data_ptr(), nnz, num_nodes,
         rowptr.data_ptr(), col.data_ptr(), work.data_ptr(), _stream())
    return CSR(rowptr, col[:nnz], num_nodes, nnz)


def segments_from_batch(batch, num_graphs):
    require_device(batch, "batch")
    if batch.dtype != torch.int64:
        raise ValueError("blgnn: batch must be int64")
    batch = batch.contiguous()
    seg = torch.empty(num_graphs + 1, dtype=torch.int32, device=batch.device)
    call("blgnn_segments_from_batch", batch.data_ptr(), batch.numel(), num_graphs,
         seg.data_ptr(), _stream())
    return seg


def _out(out, rows, cols, like):
    if out is None:
        return torch.empty((rows, cols), dtype=torch.float32, device=like.device)
    if tuple(out.shape) != (rows, cols):
        raise ValueError("blgnn: out has shape %s, expected %s" % (tuple(out.shape), (rows, cols)))
    return out


def spmm_sum(csr, x, self_coef=0.0, skip_self=False, w=None, out=None):
    xp, ldx = _mat(x, "x")
    n, d = x.shape
    if n != csr.n:
        raise ValueError("blgnn: x has %d rows, graph has %d nodes" % (n, csr.n))
    y = _out(out, n, d, x)
    yp, ldy = _mat(y, "out")
    call("blgnn_spmm_sum", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, _vec(w, "w"), xp, ldx,
         yp, ldy, n, d, float(self_coef), int(bool(skip_self)), _stream())
    return y


def spmm_sum_sp(csr, x, self_coef=0.0, skip_self=False, out_sp=None):
    """``spmm_sum`` with the result as an ``SpRows`` (the X operand of ``linear_sp``) and no fp32 copy."""
    xp, ldx = _mat(x, "x")
    n, d = x.shape
    if n != csr.n:
        raise ValueError("blgnn: x has %d rows, graph has %d nodes" % (n, csr.n))
    sp = out_sp if out_sp is not None else SpRows(n, d, x.device)
    call("blgnn_spmm_sum_sp", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, xp, ldx, n, d, float(self_coef),
         int(bool(skip_self)), *_sp_args(sp, n, d), _stream())
    return sp


def gin_sp_preferred(csr, x, N, force=False):
    """Shapes where a GIN layer runs as ``spmm_sum_sp`` + two ``linear_sp`` GEMMs instead of one fused launch: whole
    256-column blocks on both sides, a molecular (low-degree) graph, enough rows for the 256 x 256 tiles -- and a width
    the single-launch f16x3 layer does not have (d = 512: measured 76.8 vs 78.8 ms per step at configs[4] against the
    in-kernel-split GEMMs; at d = 256 the single launch stays ahead: configs[3] 32.9 vs 34.0 ms, configs[2] even)."""
    K = x.shape[1]
    if force:                                              # (tests: any batch size, any whole-block width)
        return K % 256 == 0 and N % 256 == 0 and 0 < K <= 2048 and 0 < N <= 2048 and x.stride(0) % 4 == 0 \
            and x.data_ptr() % 16 == 0
    return (N >= 512 and linear_sp_supported(x.shape[0], K, N) and linear_sp_supported(x.shape[0], N, N)
            and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and csr.nnz < 12 * max(csr.n, 1))


def gin_layer_supported(csr, x, K, N):
    """Shapes the single-launch fused GIN layer is written for (include/blgnn.h), on
    low-degree (molecular) graphs; high-degree interaction graphs take the one-workgroup-
    per-row aggregation kernel followed by the two GEMMs."""
    return (N in (64, 128, 256) and K % 4 == 0 and K <= 256 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and csr.nnz < 12 * max(csr.n, 1))


def gin_layer(csr, x, w1_kn, b1, w2_kn, b2, eps=0.0, act="identity", act_param=None,
              bn_scale=None, bn_shift=None, out=None):
    """Whole GIN NodeEmbedding layer (aggregate -> Linear -> act -> Linear -> act -> BN) in
    one kernel; the aggregated tile and the hidden activation never leave LDS."""
    xp, ldx = _mat(x, "x")
    n, K = x.shape
    N = w1_kn.shape[1]
    if w1_kn.shape[0] != K or tuple(w2_kn.shape) != (N, N):
        raise ValueError("blgnn: gin_layer weight shapes %s / %s do not match x [%d,%d]"
                         % (tuple(w1_kn.shape), tuple(w2_kn.shape), n, K))
    y = _out(out, n, N, x)
    yp, ldy = _mat(y, "out")
    call("blgnn_gin_layer", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, xp, ldx, n, K,
         _vec(w1_kn, "w1", optional=False), _vec(b1, "b1"), _vec(w2_kn, "w2", optional=False),
         _vec(b2, "b2"), N, float(eps), _act(act), _vec(act_param, "act_param"),
         _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), yp, ldy, _stream())
    return y


def pack_weight_x6(w_kn):
    """[K, N] fp32 weights -> the three-piece bf16 packing the bf16x6 kernels read."""
    wp, ldw = _mat(w_kn, "w")
    K, N = w_kn.shape
    out = torch.empty(int(_lib.load().blgnn_packed_weight_x6_elems(K, N)), dtype=torch.bfloat16,
                      device=w_kn.device)
    call("blgnn_pack_weight_x6", wp, ldw, K, N, out.data_ptr(), _stream())
    return out


def pack_weight_h3(w_kn):
    """[K, N] fp32 weights -> the scaled two-piece fp16 image the f16x3 kernels read."""
    wp, ldw = _mat(w_kn, "w")
    K, N = w_kn.shape
    out = torch.empty(int(_lib.load().blgnn_packed_weight_h3_elems(K, N)), dtype=torch.float16,
                      device=w_kn.device)
    call("blgnn_pack_weight_h3", wp, ldw, K, N, out.data_ptr(), _stream())
    return out


def pack_weight_h3g(w_kn):
    """[K, N] fp32 weights -> the tile-ordered two-piece fp16 image the f16x3 GEMM (linear) reads."""
    wp, ldw = _mat(w_kn, "w")
    K, N = w_kn.shape
    out = torch.empty(int(_lib.load().blgnn_packed_weight_h3g_elems(K, N)), dtype=torch.float16,
                      device=w_kn.device)
    call("blgnn_pack_weight_h3g", wp, ldw, K, N, out.data_ptr(), _stream())
    return out


def gin_layer_h3_supported(csr, x, K, N):
    return (N in (128, 256) and K % 4 == 0 and K <= N and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and csr.nnz < 12 * max(csr.n, 1))


def _pool_args(pool, n, N):
    """``pool`` = (row_graph int32[n], seg int32[G+1], pooled [G, N] view, kind) -> the five readout arguments."""
    if pool is None:
        return None, None, None, 0, 0
    row_graph, seg, pooled, kind = pool
    pp, ldp = _mat(pooled, "pooled")
    if pooled.shape != (seg.numel() - 1, N) or row_graph.numel() != n:
        raise ValueError("blgnn: fused readout expects pooled [G, %d] and one graph id per row" % N)
    return (_vec(row_graph, "row_graph", torch.int32, False), _vec(seg, "seg", torch.int32, False), pp, ldp,
            REDUCE_MEAN if kind == "mean" else REDUCE_SUM)


def gin_layer_h3(csr, x, w1_packed, b1, w2_packed, b2, N, eps=0.0, act="identity", act_param=None,
                 bn_scale=None, bn_shift=None, out=None, pool=None, amax_out=None, out_sp=None):
    """blgnn_gin_layer with both GEMMs on the fp16 matrix cores at fp32 accuracy (f16x3).  ``pool``: fused
    node -> graph readout of the result into a zero-initialised [G, N] view (see ``_pool_args``).  ``amax_out``
    [n]: receives the row magnitudes of the result (for a following f16x3 GEMM).  ``out_sp`` (``SpRows`` [n, 256],
    N == 256): the result additionally in the sp format ``linear_sp`` reads."""
    xp, ldx = _mat(x, "x")
    n, K = x.shape
    y = _out(out, n, N, x)
    yp, ldy = _mat(y, "out")
    call("blgnn_gin_layer_h3", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, xp, ldx, n, K,
         w1_packed.data_ptr(), _vec(b1, "b1"), w2_packed.data_ptr(), _vec(b2, "b2"), N, float(eps), _act(act),
         _vec(act_param, "act_param"), _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), yp, ldy,
         *_pool_args(pool, n, N), _vec(amax_out, "amax_out"), *_sp_args(out_sp, n, N), _stream())
    return y


def gin_layer_h3_codes_supported(csr, codes, offsets, N):
    return (N == 128 and codes is not None and codes.dtype == torch.int8 and codes.dim() == 2 and codes.shape[1] == 8
            and codes.is_contiguous() and len(offsets) - 1 <= 8 and offsets[-1] % 4 == 0 and offsets[-1] <= 64
            and csr.nnz < 12 * max(csr.n, 1))


def gin_layer_h3_codes(csr, codes, offsets, w1_packed, b1, w2_packed, b2, N, eps=0.0, act="identity", act_param=None,
                       bn_scale=None, bn_shift=None, out=None, pool=None):
    """blgnn_gin_layer_h3 on categorical input: ``codes`` int8 [n, 8] (one value index per attribute),
    ``offsets`` = first one-hot column of every attribute + the total width."""
    import ctypes
    require_device(codes, "codes")
    n = codes.shape[0]
    y = _out(out, n, N, w1_packed)
    if y.dtype != torch.float32:
        raise ValueError("blgnn: out must be float32")
    yp, ldy = _mat(y, "out")
    offs = (ctypes.c_int32 * len(offsets))(*[int(o) for o in offsets])
    call("blgnn_gin_layer_h3_codes", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, codes.data_ptr(), offs,
         len(offsets) - 1, n, w1_packed.data_ptr(), _vec(b1, "b1"), w2_packed.data_ptr(), _vec(b2, "b2"), N,
         float(eps), _act(act), _vec(act_param, "act_param"), _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"),
         yp, ldy, *_pool_args(pool, n, N), None, _stream())
    return y


def gin_layer_x6_supported(csr, x, K, N):
    return (N in (128, 256) and K % 4 == 0 and K <= 256 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0 and csr.nnz < 12 * max(csr.n, 1))


def gin_layer_x6(csr, x, w1_packed, b1, w2_packed, b2, N, eps=0.0, act="identity", act_param=None,
                 bn_scale=None, bn_shift=None, out=None):
    """blgnn_gin_layer with both GEMMs on the bf16 matrix cores at fp32 accuracy (bf16x6)."""
    xp, ldx = _mat(x, "x")
    n, K = x.shape
    y = _out(out, n, N, x)
    yp, ldy = _mat(y, "out")
    call("blgnn_gin_layer_x6", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, xp, ldx, n, K,
         w1_packed.data_ptr(), _vec(b1, "b1"), w2_packed.data_ptr(), _vec(b2, "b2"), N, float(eps), _act(act),
         _vec(act_param, "act_param"), _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), yp, ldy,
         _stream())
    return y


def molecule_tiles(sizes, max_rows=64):
    """Greedy packing of CONSECUTIVE molecules into tiles of <= max_rows rows (host side).
    Returns (tile_row [T+1], tile_mol [T+1]) as int32 numpy arrays, or None when a molecule is
    larger than a tile (such molecules must go through the per-layer kernels)."""
    import numpy as np
    sizes = np.asarray(sizes, dtype=np.int64)
    if sizes.size == 0 or sizes.max() > max_rows:
        return None
    tile_row, tile_mol, fill, row = [0], [0], 0, 0
    for g, n in enumerate(sizes.tolist()):
        if fill + n > max_rows:
            tile_row.append(row); tile_mol.append(g); fill = 0
        fill += n
        row += n
    tile_row.append(row); tile_mol.append(len(sizes))
    return np.asarray(tile_row, dtype=np.int32), np.asarray(tile_mol, dtype=np.int32)


def gin_stack(csr, x0, tile_row, tile_mol, seg, layers, pooled, pool_cols, kind="mean", row_map=None,
              y_last=None):
    """L stacked GIN layers + per-layer readout in one launch (include/blgnn.h: blgnn_gin_stack).
    ``layers``: list of dicts with w1 [K,N], b1, w2 [N,N], b2, act (name), eps, and optional
    slope / scale / shift; ``pool_cols[l]`` = column offset in ``pooled`` or -1.  Dicts that carry
    ``w1p`` / ``w2p`` (pack_weight_h3 images) and ``N`` instead of w1 / w2 select the f16x3 stack
    (blgnn_gin_stack_h3: N == 128, K0 <= 128)."""
    import ctypes
    xp, ldx0 = _mat(x0, "x0")
    K0 = x0.shape[1]
    L = len(layers)
    N = int(layers[0]["N"]) if "N" in layers[0] else int(layers[0]["w1"].shape[1])
    T = tile_row.numel() - 1

    def table(key, required=False):
        return (ctypes.c_void_p * L)(*[_vec(lay.get(key), key, optional=not required) for lay in layers])

    pp, ldp = _mat(pooled, "pooled") if pooled is not None else (None, 0)
    yp, ldy = _mat(y_last, "y_last") if y_last is not None else (None, 0)
    packed = all("w1p" in lay and "w2p" in lay for lay in layers)
    if packed:                                               # f16x3 stack: packed fp16 weight images
        w1t = (ctypes.c_void_p * L)(*[lay["w1p"].data_ptr() for lay in layers])
        w2t = (ctypes.c_void_p * L)(*[lay["w2p"].data_ptr() for lay in layers])
    else:
        w1t, w2t = table("w1", True), table("w2", True)
    call("blgnn_gin_stack_h3" if packed else "blgnn_gin_stack", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz,
         xp, ldx0, K0, N, T,
         _vec(tile_row, "tile_row", torch.int32, False), _vec(tile_mol, "tile_mol", torch.int32, False),
         _vec(seg, "seg", torch.int32, False), L, w1t, table("b1"), w2t,
         table("b2"), table("slope"), table("scale"), table("shift"),
         (ctypes.c_int32 * L)(*[_act(lay["act"]) for lay in layers]),
         (ctypes.c_float * L)(*[float(lay.get("eps", 0.0)) for lay in layers]),
         (ctypes.c_int32 * L)(*[int(c) for c in pool_cols]),
         REDUCE_MEAN if kind == "mean" else REDUCE_SUM, pp, ldp,
         _vec(row_map, "row_map", torch.int32), yp, ldy, _stream())
    return pooled


# Packed (bf16x6) images of weight matrices, keyed by the weight tensor OBJECT (weak reference, so an
# entry can never be matched by a different tensor that happens to reuse the address) and refreshed
# when the tensor is written in place (_version) or moved.
_PACKED = {}
GAT_TRANSFORM_MAX_ROWS = 8192  # above this the transform is throughput bound: tiled GEMM + scores kernel
LINEAR_X6_MIN_TILES = 256      # 128x128 output tiles; below one per CU the f32 kernels' smaller tiles win


def _packed_weight(w_kn, kind="x6"):
    key = (id(w_kn), kind)
    hit = _PACKED.get(key)
    stamp = (w_kn.data_ptr(), w_kn._version, tuple(w_kn.shape), w_kn.stride(0))
    if hit is not None and hit[0]() is w_kn and hit[1] == stamp:
        return hit[2]
    if len(_PACKED) > 512:
        for k in [k for k, v in _PACKED.items() if v[0]() is None]:
            del _PACKED[k]
    packed = pack_weight_h3g(w_kn) if kind == "h3g" else pack_weight_sp(w_kn) if kind == "sp" else pack_weight_x6(w_kn)
    _PACKED[key] = (weakref.ref(w_kn), stamp, packed)
    return packed


class SpRows:
    """An activation matrix [rows, cols] in the sp format of ``blgnn_linear_sp`` (include/blgnn.h, csrc/sp_common.hpp):
    ``planes`` fp16 [rows, 2 * cols] (per 32-column chunk: 32 high pieces, then 32 low pieces) and ``scales`` uint8
    [rows, cols / 256] (the power-of-two scale of every 256-column block as an exponent byte; rows padded to 8 bytes
    so that a GEMM reads a row's exponents with one load) -- the same bytes as fp32, split once by the producer.  ``cols(c0, c1)`` (multiples of 256) is a view: producers fill their slices independently."""

    BLOCK = 256

    def __init__(self, rows=None, cols=None, device=None, planes=None, scales=None):
        if planes is None:
            if cols % self.BLOCK != 0:
                raise ValueError("blgnn: sp matrices need a multiple of 256 columns (got %d)" % cols)
            planes = torch.empty((rows, 2 * cols), dtype=torch.float16, device=device)
            nblk = cols // self.BLOCK
            scales = torch.empty((rows, (nblk + 7) // 8 * 8), dtype=torch.uint8, device=device)[:, :nblk]
        self.planes, self.scales = planes, scales

    @property
    def shape(self):
        return (self.planes.shape[0], self.planes.shape[1] // 2)

    @property
    def device(self):
        return self.planes.device

    def cols(self, c0, c1):
        if c0 % self.BLOCK or c1 % self.BLOCK or not 0 <= c0 < c1 <= self.shape[1]:
            raise ValueError("blgnn: sp column slices start and end on multiples of 256")
        return SpRows(planes=self.planes[:, 2 * c0:2 * c1], scales=self.scales[:, c0 // self.BLOCK:c1 // self.BLOCK])


def _sp_args(sp, rows, cols):
    """-> (planes ptr, ld in fp16 elements, scales ptr, ld) for the C ABI; four nulls for None."""
    if sp is None:
        return None, 0, None, 0
    if tuple(sp.shape) != (rows, cols):
        raise ValueError("blgnn: sp matrix has shape %s, expected %s" % (tuple(sp.shape), (rows, cols)))
    pl, sc = sp.planes, sp.scales
    if pl.dtype != torch.float16 or sc.dtype != torch.uint8 or pl.stride(1) != 1 or sc.stride(1) != 1:
        raise ValueError("blgnn: sp planes must be fp16 rows, scales uint8 rows")
    require_device(pl, "sp planes")
    return pl.data_ptr(), pl.stride(0), sc.data_ptr(), sc.stride(0)


def pack_weight_sp(w_kn):
    """[K, N] fp32 weights (N % 256 == 0) -> the tile-ordered two-piece fp16 image ``linear_sp`` streams."""
    wp, ldw = _mat(w_kn, "w")
    K, N = w_kn.shape
    if N % SpRows.BLOCK != 0 or N == 0 or K == 0:
        raise ValueError("blgnn: the sp GEMM needs N %% 256 == 0 (got [%d, %d])" % (K, N))
    out = torch.empty(int(_lib.load().blgnn_packed_weight_sp_elems(K, N)), dtype=torch.float16, device=w_kn.device)
    call("blgnn_pack_weight_sp", wp, ldw, K, N, out.data_ptr(), _stream())
    return out


def split_rows_sp(x, out=None):
    """fp32 rows -> sp format (one streaming pass; producers with an sp epilogue do not need it)."""
    xp, ldx = _mat(x, "x")
    M, K = x.shape
    sp = out if out is not None else SpRows(M, K, x.device)
    call("blgnn_split_rows_sp", xp, ldx, M, K, *_sp_args(sp, M, K), _stream())
    return sp


def join_rows_sp(sp, out=None):
    """sp format -> fp32 rows."""
    M, K = sp.shape
    y = _out(out, M, K, sp.planes)
    yp, ldy = _mat(y, "out")
    call("blgnn_join_rows_sp", *_sp_args(sp, M, K), M, K, yp, ldy, _stream())
    return y


def linear_sp_supported(M, K, N):
    """Shapes ``linear_sp`` is written for: whole 256-column blocks on both sides; below ~one tile per CU the
    smaller-tile kernels win."""
    return K % 256 == 0 and 0 < K <= 2048 and N % 256 == 0 and N > 0 and M >= 32768


def linear_sp(xs, w_kn, bias=None, act="identity", act_param=None, bn_scale=None, bn_shift=None,
              rowvec=None, rowvec_w=None, out=None, out_sp=None, f32=None, addend=None, addend_rows=None, pool=None):
    """epi(X @ w_kn + bias [+ rowvec (x) rowvec_w] [+ addend[addend_rows]]) with X an ``SpRows``.  The result goes to
    fp32 rows (``out``, or a new tensor when ``f32``; default: when no ``out_sp`` is given), to ``out_sp``
    (``SpRows``), or to both.  ``addend`` fp32 [R, N] (rows picked by the int32 ``addend_rows`` [M], or row m when
    None): the other half of a K-split product.  ``pool`` = (row_graph int32 [M], seg int32 [G + 1], pooled fp32
    [G, >= N] zeroed by the caller, "sum" | "mean"): the node -> graph readout of the result, accumulated in the epilogue
    (with ``f32=False`` and no ``out_sp`` the rows themselves are not written).  -> (fp32 result or None, out_sp or None)."""
    M, K = xs.shape
    if w_kn.shape[0] != K:
        raise ValueError("blgnn: x is [%d,%d] but w is [%d,%d]" % (M, K, w_kn.shape[0], w_kn.shape[1]))
    N = w_kn.shape[1]
    if f32 is None:
        f32 = out_sp is None and pool is None
    y, yp, ldy = None, None, 0
    if f32 or out is not None:
        y = _out(out, M, N, xs.planes)
        yp, ldy = _mat(y, "out")
    ap, lda = (None, 0) if addend is None else _mat(addend, "addend")
    pool_args = (None, None, None, 0, 0)
    if pool is not None:
        row_graph, seg, pooled, kind = pool
        if row_graph.shape[0] != M or pooled.shape[1] < N or pooled.shape[0] != seg.numel() - 1:
            raise ValueError("blgnn: linear_sp pool needs row_graph [M], seg [G + 1] and pooled [G, >= N]")
        pp, ldp = _mat(pooled, "pooled")
        pool_args = (_vec(row_graph, "row_graph", torch.int32, False), _vec(seg, "seg", torch.int32, False), pp, ldp,
                     REDUCE_MEAN if kind == "mean" else REDUCE_SUM)
    if addend is not None and (addend.shape[1] != N or (addend_rows is None and addend.shape[0] < M) or
                               (addend_rows is not None and addend_rows.shape[0] != M)):
        raise ValueError("blgnn: linear_sp addend must be [R, %d] with one addend row (index) per output row" % N)
    call("blgnn_linear_sp", *_sp_args(xs, M, K), _packed_weight(w_kn, "sp").data_ptr(), yp, ldy,
         *_sp_args(out_sp, M, N), M, N, K, _vec(bias, "bias"), _vec(rowvec, "rowvec"), _vec(rowvec_w, "rowvec_w"),
         _act(act), _vec(act_param, "act_param"), _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"),
         ap, lda, _vec(addend_rows, "addend_rows", torch.int32), 0 if addend is None else addend.shape[0],
         *pool_args, _stream())
    return y, out_sp


def linear_writes_amax(x, w_kn, transform=None):
    """True when ``linear(x, w_kn, ..., transform=transform, amax_out=...)`` will fill ``amax_out``
    (the f16x3 kernel runs)."""
    if transform is not None:
        return transform == "f16x3" and x.shape[1] > 0 and w_kn.shape[1] > 0 and linear_x6_supported(x, w_kn)
    return getattr(FLAGS, "feature_transform", "bf16x6") == "f16x3" and linear_h3_preferred(x, w_kn)


def linear_h3_preferred(x, w_kn):
    """Shapes where the f16x3 GEMM (one 512-thread workgroup per CU, K step 64) beats the bf16x6 one
    (measured on MI355X, 462 k rows: K x N = 256 x 1024, 512 x 512, 1024 x 256, 1024 x 1024: 1.1-1.35x
    faster; 256 x 256 and below: its per-tile prologue / epilogue is not amortised, bf16x6 wins)."""
    return (x.shape[1] > 0 and w_kn.shape[1] > 0 and linear_x6_supported(x, w_kn)
            and x.shape[1] * w_kn.shape[1] >= 256 * 1024 and x.shape[0] >= 32768)


def linear_x6_supported(x, w_kn):
    tiles = ((x.shape[0] + 127) // 128) * ((w_kn.shape[1] + 127) // 128)
    return (x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and tiles >= LINEAR_X6_MIN_TILES)


def row_absmax(x, out=None):
    """max_k |x[m, k]| per row (the normalisation input of the f16x3 GEMM)."""
    xp, ldx = _mat(x, "x")
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    call("blgnn_row_absmax", xp, ldx, x.shape[0], x.shape[1], _vec(out, "out", optional=False), _stream())
    return out


def linear(x, w_kn, bias=None, act="identity", act_param=None, bn_scale=None, bn_shift=None,
           rowvec=None, rowvec_w=None, out=None, transform=None, row_amax=None, amax_out=None):
    """epi(x @ w_kn + bias [+ rowvec (x) rowvec_w]); w_kn is [K, N] row-major.
    ``transform``: "f16x3" / "bf16x6" (16-bit matrix cores, fp32-equivalent, large M only), "f32"
    (exact-f32 MFMA) or None = FLAGS.feature_transform.  ``row_amax`` [M]: row magnitudes of ``x`` when
    the caller already has them (f16x3 only; computed by one extra pass over x otherwise).  ``amax_out``
    [M], zero-initialised: receives the row magnitudes of the RESULT when the f16x3 kernel runs (the
    ``row_amax`` of a following GEMM on it); callers must check ``linear_writes_amax`` first."""
    xp, ldx = _mat(x, "x")
    wp, ldw = _mat(w_kn, "w")
    M, K = x.shape
    if w_kn.shape[0] != K:
        raise ValueError("blgnn: x is [%d,%d] but w is [%d,%d]" % (M, K, w_kn.shape[0], w_kn.shape[1]))
    N = w_kn.shape[1]
    y = _out(out, M, N, x)
    yp, ldy = _mat(y, "out")
    explicit = transform is not None                      # an explicit "f16x3" forces that kernel (tests, benches)
    if transform is None:
        transform = getattr(FLAGS, "feature_transform", "bf16x6")
    if transform in ("bf16x6", "f16x3") and K > 0 and N > 0 and linear_x6_supported(x, w_kn):
        # large-M GEMMs on the 16-bit matrix cores at fp32 accuracy: fp16 two-piece split (three MFMAs per
        # product) or bf16 three-piece split (six)
        if transform == "f16x3" and (explicit or linear_h3_preferred(x, w_kn)):
            if row_amax is None:
                row_amax = row_absmax(x)
            call("blgnn_linear_h3", xp, ldx, _vec(row_amax, "row_amax", optional=False),
                 _packed_weight(w_kn, "h3g").data_ptr(), yp, ldy, _vec(amax_out, "amax_out"), M, N, K,
                 _vec(bias, "bias"),
                 _vec(rowvec, "rowvec"), _vec(rowvec_w, "rowvec_w"), _act(act), _vec(act_param, "act_param"),
                 _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), _stream())
            return y
        call("blgnn_linear_x6", xp, ldx, _packed_weight(w_kn, "x6").data_ptr(), yp, ldy, M, N, K, _vec(bias, "bias"),
             _vec(rowvec, "rowvec"), _vec(rowvec_w, "rowvec_w"), _act(act), _vec(act_param, "act_param"),
             _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), _stream())
        return y
    call("blgnn_linear", xp, ldx, wp, ldw, yp, ldy, M, N, K, _vec(bias, "bias"),
         _vec(rowvec, "rowvec"), _vec(rowvec_w, "rowvec_w"), _act(act), _vec(act_param, "act_param"),
         _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), _stream())
    return y


def gat_transform(x, weight, att, h=None):
    """Row-wise front end of GATConv: h = x @ weight and both attention halves of every row.
    -> (h [n, d], s [2, n]: s[0] = h . att[:d] (target half), s[1] = h . att[d:] (source half)).
    ``h`` may be a column slice of a wider buffer.  Row-wise means shardable: a rank can transform its own
    rows and exchange [n_r, d + 2] instead of the [n_r, K] inputs."""
    n, d = x.shape[0], weight.shape[1]
    sc = torch.empty((2, n), dtype=torch.float32, device=x.device)
    attv = att.reshape(-1)
    s = _stream()
    if weight.shape[0] != x.shape[1]:
        raise ValueError("blgnn: x is [%d,%d] but w is [%d,%d]" % (n, x.shape[1], weight.shape[0], d))
    if d <= 128 and n < GAT_TRANSFORM_MAX_ROWS:
        # interaction level: x @ W and both attention halves in one split-K launch
        xp, ldx = _mat(x, "x")
        wp, ldw = _mat(weight, "w")
        h = _out(h, n, d, x)
        hp, ldh = _mat(h, "h")
        call("blgnn_gat_transform", xp, ldx, wp, ldw, _vec(attv, "att", optional=False), n, d, x.shape[1],
             hp, ldh, sc[0].data_ptr(), sc[1].data_ptr(), s)
    else:
        h = linear(x, weight, out=h)
        hp, ldh = _mat(h, "h")
        call("blgnn_gat_scores", hp, ldh, _vec(attv, "att", optional=False), n, d, sc[0].data_ptr(), sc[1].data_ptr(), s)
    return h, sc


def gat_from_transform(csr, h, sc, bias, group="src", slope=0.2, act="identity", act_param=None,
                       bn_scale=None, bn_shift=None, out=None, accumulate=False, rows=None):
    """The graph half of GATConv on an already transformed table: per-group softmax statistics and the
    weighted neighbour sum + bias + the caller's act / eval-BN epilogue.  ``rows`` (int64, device): only these target
    rows are computed (the others keep whatever ``out`` held) -- the last interaction layer of a sharded forward."""
    if group not in GAT_GROUP:
        raise ValueError("softmax_group must be 'src' or 'dst'")
    n, d = h.shape
    if tuple(sc.shape) != (2, n) or not sc.is_contiguous():
        raise ValueError("blgnn: scores must be a contiguous [2, n] tensor")
    stats = torch.empty((2, n), dtype=torch.float32, device=h.device)
    s_dst, s_src, mx, den = sc[0].data_ptr(), sc[1].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr()
    # SRC grouping on a large graph: (s_src, max, denominator) of a node also packed into one 16-byte word, so that the
    # aggregation gathers one sector per edge instead of three
    packed = torch.empty((n, 4), dtype=torch.float32, device=h.device) if (group == "src" and n >= 16384) else None
    call("blgnn_gat_softmax_stats", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, s_dst, s_src, n,
         GAT_GROUP[group], float(slope), mx, den, None if packed is None else packed.data_ptr(), _stream())
    return gat_aggregate(csr, h, sc, stats, bias, group, slope, act, act_param, bn_scale, bn_shift, out, accumulate, rows,
                         packed)


def gat_aggregate(csr, h, sc, stats, bias, group="src", slope=0.2, act="identity", act_param=None,
                  bn_scale=None, bn_shift=None, out=None, accumulate=False, rows=None, packed=None):
    """The weighted neighbour sum of GATConv (blgnn_gat_aggregate: the interaction-level gather) on a transformed table
    with its scores ``sc`` [2, n] and softmax statistics ``stats`` [2, n] (max, denominator per group; ``packed`` [n, 4]:
    the same per node in one word, as blgnn_gat_softmax_stats writes it)."""
    n, d = h.shape
    hp, ldh = _mat(h, "h")
    y = _out(out, n, d, h)
    yp, ldy = _mat(y, "out")
    call("blgnn_gat_aggregate", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, hp, ldh, sc[0].data_ptr(),
         sc[1].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), GAT_GROUP[group], float(slope), yp, ldy, n, d,
         _vec(bias, "bias"), _act(act), _vec(act_param, "act_param"), _vec(bn_scale, "bn_scale"),
         _vec(bn_shift, "bn_shift"), int(bool(accumulate)), _vec(rows, "rows", torch.int64),
         0 if rows is None else rows.numel(), None if packed is None else packed.data_ptr(), _stream())
    return y


def gat_conv(csr, x, weight, att, bias, group="src", slope=0.2, act="identity", act_param=None,
             bn_scale=None, bn_shift=None, out=None, accumulate=False, rows=None):
    """GATConv(heads=1) + the caller's act / eval-BN epilogue."""
    if group not in GAT_GROUP:
        raise ValueError("softmax_group must be 'src' or 'dst'")
    h, sc = gat_transform(x, weight, att)
    return gat_from_transform(csr, h, sc, bias, group, slope, act, act_param, bn_scale, bn_shift, out, accumulate, rows)


def gcn_conv(csr, x, weight, bias, edge_weight=None, act="identity", act_param=None,
             bn_scale=None, bn_shift=None, out=None, accumulate=False):
    h = linear(x, weight)
    n, d = h.shape
    hp, ldh = _mat(h, "h")
    dis = torch.empty(n, dtype=torch.float32, device=h.device)
    s = _stream()
    wp = _vec(edge_weight, "edge_weight")
    call("blgnn_gcn_deg_inv_sqrt", csr.rowptr.data_ptr(), csr.col.data_ptr(), wp, n,
         dis.data_ptr(), s)
    y = _out(out, n, d, h)
    yp, ldy = _mat(y, "out")
    call("blgnn_gcn_aggregate", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, wp, dis.data_ptr(), hp,
         ldh, yp, ldy, n, d, _vec(bias, "bias"), _act(act), _vec(act_param, "act_param"),
         _vec(bn_scale, "bn_scale"), _vec(bn_shift, "bn_shift"), int(bool(accumulate)), s)
    return y


def gcn_propagate(csr, h, edge_weight=None):
    """A_hat h with GCNConv's symmetric normalisation (no transform, no bias): the operator of the layer's forward
    AND, because A_hat is symmetric on the undirected graphs of this path, of its backward."""
    n, d = h.shape
    hp, ldh = _mat(h, "h")
    dis = torch.empty(n, dtype=torch.float32, device=h.device)
    s = _stream()
    wp = _vec(edge_weight, "edge_weight")
    call("blgnn_gcn_deg_inv_sqrt", csr.rowptr.data_ptr(), csr.col.data_ptr(), wp, n, dis.data_ptr(), s)
    y = torch.empty((n, d), dtype=torch.float32, device=h.device)
    call("blgnn_gcn_aggregate", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, wp, dis.data_ptr(), hp, ldh,
         y.data_ptr(), d, n, d, None, _act("identity"), None, None, None, 0, s)
    return y


def segment_reduce(seg, x, kind="mean", out=None, row_map=None):
    """scatter_mean / scatter_add over sorted segments; ``out`` may be a column
    slice of a wider buffer and ``row_map`` (int32[G]) redirects output rows."""
    xp, ldx = _mat(x, "x")
    G = seg.numel() - 1
    d = x.shape[1]
    if out is None:
        out = torch.empty((G, d), dtype=torch.float32, device=x.device)
    op, ldo = _mat(out, "out")
    if out.shape[1] != d:
        raise ValueError("blgnn: out must have %d columns" % d)
    call("blgnn_segment_reduce", _vec(seg, "seg", torch.int32, False), G, xp, ldx, d,
         REDUCE_MEAN if kind == "mean" else REDUCE_SUM, op, ldo,
         _vec(row_map, "row_map", torch.int32), _stream())
    return out


def segment_reduce_multi(seg, xs, kind="mean", row_map=None, out=None):
    """torch.cat([scatter_mean(x, batch) for x in xs], dim=1) in one launch (<= 8 inputs)."""
    import ctypes
    if not 1 <= len(xs) <= 8:
        raise ValueError("blgnn: segment_reduce_multi takes 1..8 inputs")
    G = seg.numel() - 1
    ptrs, lds, ds = [], [], []
    for i, x in enumerate(xs):
        p, ld = _mat(x, "xs[%d]" % i)
        ptrs.append(p); lds.append(ld); ds.append(x.shape[1])
    out = _out(out, G, sum(ds), xs[0])
    op, ldo = _mat(out, "out")
    n = len(xs)
    call("blgnn_segment_reduce_multi", _vec(seg, "seg", torch.int32, False), G, n,
         (ctypes.c_void_p * n)(*ptrs), (ctypes.c_int64 * n)(*lds), (ctypes.c_int32 * n)(*ds),
         REDUCE_MEAN if kind == "mean" else REDUCE_SUM, op, ldo,
         _vec(row_map, "row_map", torch.int32), _stream())
    return out


def match_pair_order(ind_list, device):
    """-> (order int32 device tensor, counts int32[4] numpy, n_le48): the pair indices of a batch grouped by the size of
    their larger graph (<= 32, <= 64, <= 128, larger), smaller pairs first inside a class, from the host-side list of node
    ranges (src/merged_graph.py:56-57); ``n_le48``: how many pairs of the second class have no graph above 48 nodes."""
    import numpy as np
    sizes = ind_list.sizes() if hasattr(ind_list, "sizes") else \
        np.asarray([int(e) - int(s) for s, e in ind_list], dtype=np.int64)
    big = np.maximum(sizes[0::2], sizes[1::2][:len(sizes[0::2])]) if len(sizes) else sizes
    cls = np.where(big <= 0, 4, np.where(big <= 32, 0, np.where(big <= 64, 1, np.where(big <= 128, 2, 3))))
    order = np.lexsort((big, cls))                        # by class, then by size (stable)
    counts = np.bincount(cls, minlength=5)[:4].astype(np.int32)
    order = order[:int(counts.sum())].astype(np.int32)
    n_le48 = int(np.count_nonzero((cls == 1) & (big <= 48)))
    return torch.as_tensor(order).to(device), np.ascontiguousarray(counts), n_le48


def gmn_match(seg, x, out=None, row_amax=None, max_nodes=0, classes=0, order=None):
    """``row_amax`` [rows]: merged with the row magnitudes of the result (atomic maximum).  ``max_nodes``: the largest
    graph of the batch if the caller knows it (host-side ind_list), so that unused size classes are not launched.
    ``classes``: bit mask of the pair size classes to process (1: both graphs <= 64 nodes, 2: 65..128, 4: larger; 0 = all)
    -- independent kernels that a caller may put on different streams.  ``order``: ``match_pair_order(...)`` of the batch."""
    xp, ldx = _mat(x, "x")
    G = seg.numel() - 1
    if G % 2:
        raise ValueError("blgnn: gmn_match needs an even number of graphs (pairs)")
    u = _out(out, x.shape[0], x.shape[1], x)
    up, ldu = _mat(u, "out")
    call("blgnn_gmn_match", _vec(seg, "seg", torch.int32, False), G // 2, xp, ldx, up, ldu,
         x.shape[1], _vec(row_amax, "row_amax"), int(max_nodes), int(classes),
         None if order is None else _vec(order[0], "order", torch.int32, False),
         None if order is None else order[1].ctypes.data, _stream())
    return u


# 48-row chunk images for the <= 64-node pairs that fit (four wavefronts per CU instead of three): 7 % faster on a launch
# of equal 45 x 45 pairs; configs[2] A/B in one process pair, ms/step: 14.00 / 14.10 without, 13.95 / 13.88 with.
# BLGNN_MATCH_SP48=0 runs the whole class with 64-row images.
_MATCH_SP48 = os.environ.get("BLGNN_MATCH_SP48", "1") != "0"


def gmn_match_sp(seg, x_sp, out_sp, order, row_map=None, x=None, max_nodes=0, part="all"):
    """The attention on sp planes with an sp result (``SpRows`` [n, 256] both): pairs of two graphs <= 64 nodes by
    ``blgnn_gmn_match_sp``; any larger pairs of the batch through the fp32 kernels on ``x`` (fp32 rows, required then)
    and a split of just their rows.  ``order`` = ``match_pair_order(...)``; ``row_map`` int32 [n]: plane row of ``x_sp``
    per merged atom (x kept on the distinct molecules), None = identity.  ``part``: "small" / "big" run only one of the
    two groups (they are independent: a caller may put them on different streams).  -> number of larger pairs."""
    n, d = out_sp.shape
    G = seg.numel() - 1
    if G % 2:
        raise ValueError("blgnn: gmn_match needs an even number of graphs (pairs)")
    xr = x_sp.shape[0]
    if x_sp.shape[1] != d or (row_map is None and xr != n):
        raise ValueError("blgnn: gmn_match_sp operands differ in shape")
    counts = order[1]
    n_big = int(counts[2]) + int(counts[3])
    if part in ("all", "small"):
        call("blgnn_gmn_match_sp", _vec(seg, "seg", torch.int32, False), G // 2, *_sp_args(x_sp, xr, d),
             _vec(row_map, "row_map", torch.int32), xr, *_sp_args(out_sp, n, d), d,
             _vec(order[0], "order", torch.int32, False), counts.ctypes.data,
             int(order[2]) if (len(order) > 2 and _MATCH_SP48) else 0, _stream())
    if n_big and part in ("all", "big"):
        if x is None:
            raise ValueError("blgnn: this batch has pairs above 64 nodes: gmn_match_sp needs the fp32 rows of x for them")
        u = gmn_match(seg, x, max_nodes=max_nodes, classes=6, order=order)
        big = order[0][int(counts[0]) + int(counts[1]):]
        xp, ldx = _mat(u, "u")
        call("blgnn_split_rows_sp_pairs", xp, ldx, _vec(seg, "seg", torch.int32, False), big.data_ptr(), n_big, d,
             *_sp_args(out_sp, n, d), _stream())
    return n_big


def edge_relu_sum(csr, p, q, out=None, row_amax=None, out_sp=None):
    """``row_amax`` [n]: merged with the row maxima of the result (atomic maximum).  ``out_sp`` (``SpRows`` [n, d],
    d % 256 == 0): the result goes there in the sp format INSTEAD of fp32 rows (returns ``out_sp``)."""
    pp, ldp = _mat(p, "p")
    qp, ldq = _mat(q, "q")
    n, d = p.shape
    if out_sp is not None:
        call("blgnn_edge_relu_sum", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, pp, ldp, qp, ldq, None,
             0, n, d, None, *_sp_args(out_sp, n, d), _stream())
        return out_sp
    y = _out(out, n, d, p)
    yp, ldy = _mat(y, "out")
    call("blgnn_edge_relu_sum", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, pp, ldp, qp, ldq, yp,
         ldy, n, d, _vec(row_amax, "row_amax"), None, 0, None, 0, _stream())
    return y


def row_degree(csr):
    if csr._deg is None:
        deg = torch.empty(csr.n, dtype=torch.float32, device=csr.rowptr.device)
        call("blgnn_row_degree", csr.rowptr.data_ptr(), csr.n, deg.data_ptr(), _stream())
        csr._deg = deg
    return csr._deg


def pair_gather_normalize(z, ids1, ids2, eps=1e-12):
    zp, ldz = _mat(z, "z")
    B, d = ids1.numel(), z.shape[1]
    out = torch.empty((B, 2 * d), dtype=torch.float32, device=z.device)
    call("blgnn_pair_gather_normalize", zp, ldz, _vec(ids1, "ids1", torch.int64, False),
         _vec(ids2, "ids2", torch.int64, False), B, d, float(eps), out.data_ptr(), 2 * d, _stream())
    return out


def link_pred_head(z, ids1, ids2, weights_kn, biases, hidden_act="relu", final_act="identity", eps=1e-12):
    """normalize + gather + concat + MLP (+ sigmoid) of the pair scorer in one launch.
    ``weights_kn``: list of [K_l, N_l] row-major tensors; ``biases``: list of [N_l] or None."""
    import ctypes
    zp, ldz = _mat(z, "z")
    B, d, L = ids1.numel(), z.shape[1], len(weights_kn)
    dims = [2 * d] + [int(w.shape[1]) for w in weights_kn]
    for l, w in enumerate(weights_kn):
        if w.shape[0] != dims[l]:
            raise ValueError("blgnn: link_pred_head layer %d expects %d inputs, weight has %d" % (l, dims[l], w.shape[0]))
    out = torch.empty((B, dims[-1]), dtype=torch.float32, device=z.device)
    wp = (ctypes.c_void_p * L)(*[_vec(w, "w[%d]" % l, optional=False) for l, w in enumerate(weights_kn)])
    bp = (ctypes.c_void_p * L)(*[_vec(b, "b[%d]" % l) for l, b in enumerate(biases)])
    call("blgnn_link_pred_head", zp, ldz, _vec(ids1, "ids1", torch.int64, False),
         _vec(ids2, "ids2", torch.int64, False), B, d, float(eps), L, wp, bp, (ctypes.c_int32 * (L + 1))(*dims),
         _act(hidden_act), _act(final_act), out.data_ptr(), dims[-1], _stream())
    return out


def l2_normalize_rows(x, eps=1e-12, out=None):
    xp, ldx = _mat(x, "x")
    y = _out(out, x.shape[0], x.shape[1], x)
    yp, ldy = _mat(y, "out")
    call("blgnn_l2_normalize_rows", xp, ldx, yp, ldy, x.shape[0], x.shape[1], float(eps), _stream())
    return y


def scatter_rows(src, row_map, dst):
    sp, lds = _mat(src, "src")
    dp, ldd = _mat(dst, "dst")
    call("blgnn_scatter_rows", sp, lds, _vec(row_map, "row_map", torch.int32, False), dp, ldd,
         src.shape[0], src.shape[1], _stream())
    return dst


def expand_rows(src, row_map, out=None):
    """out[i] = src[row_map[i]] for 2-D tensors of any dtype (rows moved as bytes): the replication step of a pair
    batch whose pair-independent layers ran on the distinct molecules (include/blgnn.h: blgnn_expand_rows)."""
    require_device(src, "src")
    if src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("blgnn: expand_rows needs 2-D rows with unit column stride")
    n = int(row_map.shape[0])
    if out is None:
        out = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
    if out.dtype != src.dtype or tuple(out.shape) != (n, src.shape[1]) or out.stride(1) != 1:
        raise ValueError("blgnn: expand_rows output must be [%d, %d] %s rows" % (n, src.shape[1], src.dtype))
    item = src.element_size()
    call("blgnn_expand_rows", src.data_ptr(), src.stride(0) * item, _vec(row_map, "row_map", torch.int32, False), n,
         out.data_ptr(), out.stride(0) * item, src.shape[1] * item, _stream())
    return out


def expand_rows_sp(src, row_map, out):
    """``expand_rows`` on an ``SpRows`` (column slices allowed): piece planes and scale bytes."""
    expand_rows(src.planes, row_map, out=out.planes)
    expand_rows(src.scales, row_map, out=out.scales)
    return out


# ---- backward pass / train mode (include/blgnn.h "backward pass" block) -------------------------------------

_COL_MODES = {"sum": 0, "dot": 1, "sqdev": 2, "dot_norm": 3, "dot_neg": 4}


def col_reduce(a, mode="sum", b=None, mean=None, rstd=None, scale=1.0, out=None):
    """Deterministic column reduction over all rows: sum a | sum a*b | sum (a - mean)^2 |
    sum a * (b - mean) * rstd | sum a * min(b, 0), times ``scale``."""
    ap, lda = _mat(a, "a")
    M, N = a.shape
    bp, ldb = (None, 0) if b is None else _mat(b, "b")
    if b is not None and tuple(b.shape) != (M, N):
        raise ValueError("blgnn: col_reduce operands differ in shape")
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    scratch = torch.empty(max(int(_lib.load().blgnn_col_reduce_scratch_elems(M, N)), 1), dtype=torch.float32, device=a.device)
    call("blgnn_col_reduce", ap, lda, bp, ldb, _vec(mean, "mean"), _vec(rstd, "rstd"), M, N, _COL_MODES[mode], float(scale),
         scratch.data_ptr(), _vec(out, "out", optional=False), _stream())
    return out


def bn_finalize(mean, var, eps, momentum=0.0, unbias=1.0, running_mean=None, running_var=None):
    """rstd = 1 / sqrt(var + eps) and (optionally) BatchNorm's running-statistics update, one launch.  -> rstd [N]"""
    N = var.shape[0]
    rstd = torch.empty(N, dtype=torch.float32, device=var.device)
    call("blgnn_bn_finalize", _vec(mean, "mean"), _vec(var, "var", optional=False), N, float(eps), float(momentum),
         float(unbias), rstd.data_ptr(), _vec(running_mean, "running_mean"), _vec(running_var, "running_var"), _stream())
    return rstd


def bn_apply(x, mean, rstd, gamma=None, beta=None, out=None, act="identity"):
    xp, ldx = _mat(x, "x")
    y = _out(out, x.shape[0], x.shape[1], x)
    yp, ldy = _mat(y, "out")
    call("blgnn_bn_apply", xp, ldx, _vec(mean, "mean", optional=False), _vec(rstd, "rstd", optional=False),
         _vec(gamma, "gamma"), _vec(beta, "beta"), x.shape[0], x.shape[1], _act(act), yp, ldy, _stream())
    return y


def bn_backward(dy, x, mean, rstd, gamma, dgamma, dbeta, out=None, count=None):
    dyp, lddy = _mat(dy, "dy")
    xp, ldx = _mat(x, "x")
    dx = _out(out, x.shape[0], x.shape[1], x)
    dxp, lddx = _mat(dx, "out")
    call("blgnn_bn_backward", dyp, lddy, xp, ldx, _vec(mean, "mean", optional=False), _vec(rstd, "rstd", optional=False),
         _vec(gamma, "gamma"), _vec(dgamma, "dgamma", optional=False), _vec(dbeta, "dbeta", optional=False),
         x.shape[0], x.shape[1], int(round(count)) if count is not None else x.shape[0], dxp, lddx, _stream())
    return dx


def act_forward(x, act, slope=None, out=None):
    """y = act(x) as its own launch (per-column PReLU slopes): the training path keeps the pre-activation."""
    xp, ldx = _mat(x, "x")
    y = _out(out, x.shape[0], x.shape[1], x)
    yp, ldy = _mat(y, "out")
    call("blgnn_act_forward", xp, ldx, _act(act), _vec(slope, "slope"), x.shape[0], x.shape[1], yp, ldy, _stream())
    return y


def act_backward(dy, y, act, slope=None, out=None):
    dyp, lddy = _mat(dy, "dy")
    yp, ldy = _mat(y, "y")
    dx = _out(out, y.shape[0], y.shape[1], y)
    dxp, lddx = _mat(dx, "out")
    call("blgnn_act_backward", dyp, lddy, yp, ldy, _act(act), _vec(slope, "slope"), y.shape[0], y.shape[1], dxp, lddx,
         _stream())
    return dx


def transpose(x, out=None):
    xp, ldx = _mat(x, "x")
    M, N = x.shape
    y = _out(out, N, M, x)
    yp, ldy = _mat(y, "out")
    call("blgnn_transpose", xp, ldx, M, N, yp, ldy, _stream())
    return y


def segment_broadcast(seg, dp, n_rows, kind="mean", row_map=None, out=None, accumulate=False):
    """Gradient of ``segment_reduce``: rows of segment g receive dp[g] (/ count for mean)."""
    dpp, lddp = _mat(dp, "dp")
    G, d = seg.numel() - 1, dp.shape[1]
    dx = _out(out, n_rows, d, dp)
    dxp, lddx = _mat(dx, "out")
    call("blgnn_segment_broadcast", _vec(seg, "seg", torch.int32, False), G, dpp, lddp,
         _vec(row_map, "row_map", torch.int32), d, REDUCE_MEAN if kind == "mean" else REDUCE_SUM, dxp, lddx,
         int(bool(accumulate)), _stream())
    return dx


def l2_normalize_backward(z, dzn, eps=1e-12, out=None):
    zp, ldz = _mat(z, "z")
    gp, ldg = _mat(dzn, "dzn")
    dz = _out(out, z.shape[0], z.shape[1], z)
    dzp, lddz = _mat(dz, "out")
    call("blgnn_l2_normalize_backward", zp, ldz, gp, ldg, z.shape[0], z.shape[1], float(eps), dzp, lddz, _stream())
    return dz


def pair_scatter_add(dcat, col0, ids, dz):
    cp, ldc = _mat(dcat, "dcat")
    dzp, lddz = _mat(dz, "dz")
    call("blgnn_pair_scatter_add", cp, ldc, int(col0), _vec(ids, "ids", torch.int64, False), ids.numel(), dz.shape[1],
         dzp, lddz, _stream())
    return dz


def gat_forward_saved(csr, x, weight, att, bias, group="src", slope=0.2, act="identity", out=None, accumulate=False):
    """GATConv forward that also returns what its backward needs: -> (y, h, sc [2, n], stats [2, n]).
    ``accumulate``: add into ``out`` before the activation (sum over edge types)."""
    if group not in GAT_GROUP:
        raise ValueError("softmax_group must be 'src' or 'dst'")
    h, sc = gat_transform(x, weight, att)
    n, d = h.shape
    stats = torch.empty((2, n), dtype=torch.float32, device=h.device)
    s = _stream()
    hp, ldh = _mat(h, "h")
    call("blgnn_gat_softmax_stats", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, sc[0].data_ptr(), sc[1].data_ptr(), n,
         GAT_GROUP[group], float(slope), stats[0].data_ptr(), stats[1].data_ptr(), None, s)
    y = _out(out, n, d, h)
    yp, ldy = _mat(y, "out")
    call("blgnn_gat_aggregate", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, hp, ldh, sc[0].data_ptr(),
         sc[1].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), GAT_GROUP[group], float(slope), yp, ldy, n, d,
         _vec(bias, "bias"), _act(act), None, None, None, int(bool(accumulate)), None, 0, None, s)
    return y, h, sc, stats


def gat_backward(csr, h, sc, stats, dz, att, group="src", slope=0.2):
    """-> (dh [n, d], ds [2, n]): gradient of the graph half of GATConv w.r.t. the transformed table and the two
    attention halves (see include/blgnn.h: blgnn_gat_backward_*)."""
    n, d = h.shape
    hp, ldh = _mat(h, "h")
    gp, ldg = _mat(dz, "dz")
    ds = torch.zeros((2, n), dtype=torch.float32, device=h.device)
    dh = torch.empty((n, d), dtype=torch.float32, device=h.device)
    s = _stream()
    call("blgnn_gat_backward_scores", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, hp, ldh, gp, ldg, sc[0].data_ptr(),
         sc[1].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), n, d, GAT_GROUP[group], float(slope), ds[0].data_ptr(),
         ds[1].data_ptr(), s)
    call("blgnn_gat_backward_features", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, gp, ldg, sc[0].data_ptr(),
         sc[1].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ds[0].data_ptr(), ds[1].data_ptr(),
         _vec(att.reshape(-1).contiguous(), "att", optional=False), n, d, GAT_GROUP[group], float(slope), dh.data_ptr(), d, s)
    return dh, ds


def edge_relu_sum_backward(csr, p, q, ds):
    """-> (dp, dq) of ``edge_relu_sum`` on a symmetric CSR."""
    pp, ldp = _mat(p, "p")
    qp, ldq = _mat(q, "q")
    gp, ldg = _mat(ds, "ds")
    n, d = p.shape
    dp = torch.empty((n, d), dtype=torch.float32, device=p.device)
    dq = torch.empty((n, d), dtype=torch.float32, device=p.device)
    call("blgnn_edge_relu_sum_backward", csr.rowptr.data_ptr(), csr.col.data_ptr(), csr.nnz, pp, ldp, qp, ldq, gp, ldg,
         n, d, dp.data_ptr(), d, dq.data_ptr(), d, _stream())
    return dp, dq


def gmn_match_backward(seg, x, du):
    """dx of ``gmn_match`` (pairs = consecutive graphs of ``seg``)."""
    xp, ldx = _mat(x, "x")
    gp, ldg = _mat(du, "du")
    n, d = x.shape
    P = (seg.numel() - 1) // 2
    sizes = (seg[1:] - seg[:-1]).to(torch.int64)                      # device-side index arithmetic (plumbing)
    cells = sizes[0::2] * sizes[1::2]
    cell_off = torch.zeros(P + 1, dtype=torch.int64, device=x.device)
    torch.cumsum(cells, 0, out=cell_off[1:])
    total_cells = int(cell_off[-1])
    cell_scratch = torch.empty(max(4 * total_cells, 1), dtype=torch.float32, device=x.device)
    row_scratch = torch.empty(max(3 * n, 1), dtype=torch.float32, device=x.device)
    dx = torch.empty((n, d), dtype=torch.float32, device=x.device)
    call("blgnn_gmn_match_backward", _vec(seg, "seg", torch.int32, False), P, xp, ldx, gp, ldg, cell_off.data_ptr(),
         total_cells, cell_scratch.data_ptr(), row_scratch.data_ptr(), dx.data_ptr(), d, d, n, _stream())
    return dx


def grad_weight(g, x):
    """dw[o, i] = sum_m g[m, o] * x[m, i] over all rows (deterministic chunked reduction on the matrix cores)."""
    gp, ldg = _mat(g, "g")
    xp, ldx = _mat(x, "x")
    M, O = g.shape
    I = x.shape[1]
    if x.shape[0] != M:
        raise ValueError("blgnn: grad_weight operands differ in rows")
    scratch = torch.empty(max(int(_lib.load().blgnn_grad_weight_scratch_elems(M, O, I)), 1), dtype=torch.float32,
                          device=g.device)
    dw = torch.empty((O, I), dtype=torch.float32, device=g.device)
    call("blgnn_grad_weight", gp, ldg, xp, ldx, M, O, I, scratch.data_ptr(), dw.data_ptr(), _stream())
    return dw


def gru_gates(gi, gh, h, out=None):
    """torch.nn.GRUCell's gate arithmetic on the two [n, 3D] gate tables and the state h [n, D]."""
    gip, ldgi = _mat(gi, "gi")
    ghp, ldgh = _mat(gh, "gh")
    hp, ldh = _mat(h, "h")
    n, D = h.shape
    if gi.shape != (n, 3 * D) or gh.shape != (n, 3 * D):
        raise ValueError("blgnn: gru_gates expects [n, 3D] gate tables")
    y = _out(out, n, D, h)
    yp, ldy = _mat(y, "out")
    call("blgnn_gru_gates", gip, ldgi, ghp, ldgh, hp, ldh, n, D, yp, ldy, _stream())
    return y


def gru_gates_backward(gi, gh, h, g):
    """-> (dgi [n, 3D], dgh [n, 3D], dh [n, D]) of ``gru_gates`` for the output gradient ``g``."""
    gip, ldgi = _mat(gi, "gi")
    ghp, ldgh = _mat(gh, "gh")
    hp, ldh = _mat(h, "h")
    gp, ldg = _mat(g, "g")
    n, D = h.shape
    if gi.shape != (n, 3 * D) or gh.shape != (n, 3 * D) or g.shape != (n, D):
        raise ValueError("blgnn: gru_gates_backward expects [n, 3D] gate tables and [n, D] state / gradient")
    dgi, dgh = torch.empty_like(gi), torch.empty_like(gh)
    dh = torch.empty((n, D), dtype=torch.float32, device=h.device)
    call("blgnn_gru_gates_backward", gip, ldgi, ghp, ldgh, hp, ldh, gp, ldg, n, D, dgi.data_ptr(), dgi.stride(0),
         dgh.data_ptr(), dgh.stride(0), dh.data_ptr(), dh.stride(0), _stream())
    return dgi, dgh, dh


def hadamard(a, b, out=None):
    ap, lda = _mat(a, "a")
    bp, ldb = _mat(b, "b")
    if a.shape != b.shape:
        raise ValueError("blgnn: hadamard operands differ in shape")
    y = _out(out, a.shape[0], a.shape[1], a)
    yp, ldy = _mat(y, "out")
    call("blgnn_hadamard", ap, lda, bp, ldb, a.shape[0], a.shape[1], yp, ldy, _stream())
    return y
