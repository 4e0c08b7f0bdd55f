"""GPU parity of the pre-split ("sp") GEMM pipeline of the GMN propagator (include/blgnn.h: blgnn_linear_sp,
blgnn_split_rows_sp / blgnn_join_rows_sp, the sp epilogues of blgnn_edge_relu_sum and blgnn_gin_layer_h3):
every piece against the fp64 oracle at the 1e-5 bar, then the whole GMN models at d=256 / d=512 with the
pipeline forced on (reference: model/layers_gmn.py:34-84, model/layers_util.py:44-57)."""

import numpy as np
import pytest
import torch

import blgnn
import oracle
from blgnn import ops
from helpers import TOL, max_err
from test_gpu_ops import molecule_graph

pytestmark = pytest.mark.gpu


def _row_rel(got, ref):
    """largest error of any row relative to that row's largest reference magnitude"""
    ref = ref.double()
    err = (got.detach().cpu().double() - ref).abs().amax(dim=1)
    scale = ref.abs().amax(dim=1)
    live = scale > 0
    return float((err[live] / scale[live]).max()) if bool(live.any()) else 0.0


def test_split_join_round_trip(device):
    """fp32 -> two fp16 pieces + block scale -> fp32: 22 significand bits relative to the block maximum, magnitudes
    from 1e-25 to 1e25 (the power-of-two scale is clamped to 2^+-95, h3_common.hpp: pow2_scale), zero blocks, blocks
    decades apart inside a row, a column slice of a wider sp matrix"""
    M, K = 1000, 768
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g) * torch.logspace(-25, 25, M).view(-1, 1)
    x[:, 256:512] *= 1e-6
    x[7] = 0.0
    x[9, 512:] = 0.0
    x[::5] *= torch.logspace(-5, 0, K).view(1, -1)
    xd = x.to(device)
    sp = ops.split_rows_sp(xd)
    back = ops.join_rows_sp(sp).cpu()
    blocks = x.view(M, 3, 256)
    err = (back.view(M, 3, 256).double() - blocks.double()).abs().amax(dim=2)
    bmax = blocks.abs().amax(dim=2).double()
    assert bool((err <= bmax * 2.0 ** -21).all())
    assert float(back[7].abs().max()) == 0.0 and float(back[9, 512:].abs().max()) == 0.0
    # producers fill column slices of one wide matrix independently
    wide = ops.SpRows(M, 4 * 256, device)
    ops.split_rows_sp(xd[:, 256:512].contiguous(), out=wide.cols(512, 768))
    assert torch.equal(ops.join_rows_sp(wide.cols(512, 768)).cpu(), back[:, 256:512])
    with pytest.raises(ValueError):
        wide.cols(100, 356)


@pytest.mark.parametrize("M,K,N,act", [(300, 256, 256, "identity"), (1000, 512, 512, "relu"), (515, 1024, 256, "relu"),
                                       (257, 256, 1024, "tanh"), (64, 2048, 256, "prelu")])
def test_linear_sp(device, M, K, N, act):
    """epi(X @ W + bias + rowvec (x) rowvec_w) from sp operands == fp64, as fp32 rows AND as sp output; ragged M,
    every activation, eval-BN affine"""
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    b, rv, rw = torch.randn(N, generator=g) * 0.1, torch.rand(M, generator=g) * 3, torch.randn(N, generator=g) * 0.1
    slope, scale, shift = torch.rand(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = lambda t: t.double()
    ref = D(x) @ D(w) + D(b) + D(rv).view(-1, 1) * D(rw).view(1, -1)
    ref = oracle.apply_act(ref, act, D(slope)) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    xs = ops.split_rows_sp(dv(x))
    out_sp = ops.SpRows(M, N, device)
    y, _ = ops.linear_sp(xs, dv(w), dv(b), act=act, act_param=dv(slope), bn_scale=dv(scale), bn_shift=dv(shift),
                         rowvec=dv(rv), rowvec_w=dv(rw), out_sp=out_sp, f32=True)
    assert max_err(y, ref) < TOL
    assert max_err(ops.join_rows_sp(out_sp), ref) < TOL
    # fp32 only, into a column slice of a wider buffer
    widey = torch.zeros(M, N + 8, device=device)
    ops.linear_sp(xs, dv(w), dv(b), act=act, act_param=dv(slope), bn_scale=dv(scale), bn_shift=dv(shift),
                  rowvec=dv(rv), rowvec_w=dv(rw), out=widey[:, 4:4 + N])
    assert torch.equal(widey[:, 4:4 + N], y) and float(widey[:, :4].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,N,act,mapped", [(700, 256, 1024, "relu", True), (515, 512, 256, "identity", True),
                                              (300, 256, 512, "relu", False), (257, 256, 256, "tanh", True)])
def test_linear_sp_addend(device, M, K, N, act, mapped):
    """the K-split epilogue: epi(X @ W + bias + rowvec (x) rowvec_w + addend[rows]) == fp64 with the addend gathered
    by a row map (few distinct rows, repeated and unordered) or row for row; addend rows wider than N (a column
    slice); sp and fp32 outputs; ragged M"""
    g = torch.Generator().manual_seed(M + K + N + 1)
    R = 90 if mapped else M
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    b, rv, rw = torch.randn(N, generator=g) * 0.1, torch.rand(M, generator=g) * 3, torch.randn(N, generator=g) * 0.1
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    addw = torch.randn(R, N + 12, generator=g) * 2
    rows = torch.randint(0, R, (M,), generator=g).to(torch.int32) if mapped else None
    D = lambda t: t.double()
    add = D(addw[:, 4:4 + N])
    ref = D(x) @ D(w) + D(b) + D(rv).view(-1, 1) * D(rw).view(1, -1) + (add[rows.long()] if mapped else add)
    ref = oracle.apply_act(ref, act, None) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    xs = ops.split_rows_sp(dv(x))
    out_sp = ops.SpRows(M, N, device)
    y, _ = ops.linear_sp(xs, dv(w), dv(b), act=act, bn_scale=dv(scale), bn_shift=dv(shift), rowvec=dv(rv), rowvec_w=dv(rw),
                         out_sp=out_sp, f32=True, addend=dv(addw)[:, 4:4 + N], addend_rows=dv(rows) if mapped else None)
    assert max_err(y, ref) < TOL
    assert max_err(ops.join_rows_sp(out_sp), ref) < TOL
    with pytest.raises(ValueError):                                   # one addend row (index) per output row
        ops.linear_sp(xs, dv(w), addend=dv(addw)[:M // 2, 4:4 + N].contiguous() if not mapped else dv(addw)[:, 4:4 + N],
                      addend_rows=dv(rows)[:-1] if mapped else None)


@pytest.mark.parametrize("M,K,N,act,kind,rows", [(1500, 512, 1024, "relu", "mean", False), (700, 256, 256, "identity", "sum", True),
                                                 (257, 1024, 512, "relu", "mean", True)])
def test_linear_sp_pooled_readout(device, M, K, N, act, kind, rows):
    """the readout epilogue: pooled[g] = mean / sum over the rows of graph g of epi(X @ W + bias + rowvec (x) rowvec_w)
    == fp64; graphs of 1..70 rows (several per 32-row staging pass, and graphs that span passes, waves and tiles), an
    empty graph, ragged M; with the fp32 rows written as well, or not written at all"""
    g = torch.Generator().manual_seed(M + K + N + 2)
    sizes = []
    while sum(sizes) < M:
        sizes.append(int(torch.randint(1, 71, (1,), generator=g)))
    sizes[-1] -= sum(sizes) - M
    sizes.insert(3, 0)                                                # an empty graph: its pooled row stays 0
    seg = torch.tensor([0] + sizes).cumsum(0).to(torch.int32)
    G = len(sizes)
    row_graph = torch.repeat_interleave(torch.arange(G), torch.tensor(sizes)).to(torch.int32)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    b, rv, rw = torch.randn(N, generator=g) * 0.1, torch.rand(M, generator=g) * 3, torch.randn(N, generator=g) * 0.1
    D = lambda t: t.double()
    full = oracle.apply_act(D(x) @ D(w) + D(b) + D(rv).view(-1, 1) * D(rw).view(1, -1), act, None)
    ref = torch.zeros(G, N, dtype=torch.float64).index_add_(0, row_graph.long(), full)
    if kind == "mean":
        ref = ref / torch.tensor(sizes).clamp(min=1).view(-1, 1)
    dv = lambda t: t.to(device)
    pooled = torch.zeros(G, N + 4, device=device)
    y, _ = ops.linear_sp(ops.split_rows_sp(dv(x)), dv(w), dv(b), act=act, rowvec=dv(rv), rowvec_w=dv(rw), f32=rows,
                         pool=(dv(row_graph), dv(seg), pooled[:, :N], kind))
    assert max_err(pooled[:, :N], ref) < TOL
    assert float(pooled[:, N:].abs().max()) == 0.0 and float(pooled[3].abs().max()) == 0.0
    if rows:
        assert max_err(y, full) < TOL
    else:
        assert y is None
    with pytest.raises(NotImplementedError):                          # no sp output next to the readout
        ops.linear_sp(ops.split_rows_sp(dv(x)), dv(w), out_sp=ops.SpRows(M, N, device),
                      pool=(dv(row_graph), dv(seg), pooled[:, :N], kind))


@pytest.mark.parametrize("xs,ws", [(1e-6, 1.0), (3e4, 1.0), (1.0, 1e-5), (1.0, 300.0), (1e5, 1e-4), (1e-20, 1e10)])
def test_linear_sp_range(device, xs, ws):
    """block scales inside one accumulator: tiny / huge activations and weights, rows and blocks of very different
    magnitude in one tile, zero blocks in front of, between and behind live ones, six decades inside a block"""
    M, N, K = 1200, 256, 1024
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g) * xs
    x[::3] *= torch.logspace(-6, 0, K).view(1, -1)
    x[1::7] *= 1e-4
    x[2::11, 256:512] *= 1e-9                      # a block nine decades below its neighbours
    x[3::13, :256] = 0.0                           # zero block first
    x[4::13, 256:768] = 0.0                        # zero blocks in the middle
    x[6::13, 768:] = 0.0                           # zero block last
    x[5] = 0.0
    w = torch.randn(K, N, generator=g) / K ** 0.5 * ws
    ref = x.double() @ w.double()
    y, _ = ops.linear_sp(ops.split_rows_sp(x.to(device)), w.to(device))
    assert bool(torch.isfinite(y).all())
    assert _row_rel(y, ref) < TOL
    assert float(y[5].abs().max()) == 0.0


def test_linear_sp_unsupported_shapes_raise(device):
    x = ops.split_rows_sp(torch.randn(64, 256, device=device))
    with pytest.raises(ValueError):
        ops.pack_weight_sp(torch.randn(256, 192, device=device))
    with pytest.raises(ValueError):
        ops.SpRows(8, 300, device)
    with pytest.raises(ValueError):
        ops.linear_sp(x, torch.randn(512, 256, device=device))


@pytest.mark.parametrize("d", [256, 512, 1024])
def test_edge_relu_sum_sp(device, d):
    """sum_e relu(P_i + Q_j) written straight into the sp format == the fp32 result == oracle"""
    ei, _, n = molecule_graph(41, 17)
    g = torch.Generator().manual_seed(d)
    p, q = torch.randn(n, d, generator=g), torch.randn(n, d, generator=g)
    p[3] = -100.0                                   # a row whose relus are all zero
    row, col = ei[1], ei[0]
    ref = torch.zeros(n, d, dtype=torch.float64).index_add_(0, row, torch.relu(p.double()[row] + q.double()[col]))
    csr = ops.csr_from_edge_index(ei.to(device), n)
    wide = ops.SpRows(n, 2 * d, device)
    ops.edge_relu_sum(csr, p.to(device), q.to(device), out_sp=wide.cols(d, 2 * d))
    got = ops.join_rows_sp(wide.cols(d, 2 * d))
    assert max_err(got, ref) < TOL
    plain = ops.edge_relu_sum(csr, p.to(device), q.to(device))
    assert _row_rel(got, plain.cpu()) < 2.0 ** -21


def test_gin_layer_h3_sp_output(device):
    """the fused GIN layer (N = 256) writes its result as fp32 rows and, from the same registers, as sp planes"""
    ei, _, n = molecule_graph(53, 19)
    K = N = 256
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    wide = ops.SpRows(n, 4 * N, device)
    y = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.1, act="relu", out_sp=wide.cols(0, N))
    ref = oracle.apply_act(oracle.gin_conv(x.double(), ei, w1.double(), b1.double(), w2.double(), b2.double(), "relu", eps=0.1),
                           "relu", None)
    assert max_err(y, ref) < TOL
    assert _row_rel(ops.join_rows_sp(wide.cols(0, N)), y.cpu()) < 2.0 ** -21


@pytest.mark.parametrize("d,shape,dataset,edge_types,pairs", [(256, "drugcombo", "drugcombo", 2, 260), (512, "atoms40", "drugbank", 1, 120)])
def test_gmn_model_with_sp_pipeline_forced(device, d, shape, dataset, edge_types, pairs):
    """whole lower_level_gmn model (GIN + GMNPropagator x3 + readout + head) with the sp pipeline forced on at a size
    the oracle finishes in seconds: every activation against fp64; d=512 takes the unfused GIN path, so x is
    split by the streaming pass instead of the layer kernel"""
    from test_gpu_configs import _gmn_model_vs_oracle
    import blgnn.config as cfg
    orig = cfg.make_flags

    def forced(*a, **k):
        k.setdefault("gmn_sp", "force")
        f = orig(*a, **k)
        f.gin_sp = "force"            # GIN layers with K, N on 256-column blocks: aggregation into sp + two pre-split GEMMs
        return f

    blgnn.make_flags = forced
    try:
        preds = _gmn_model_vs_oracle(device, d, 120, 1500, pairs, 61 + d, dataset, edge_types, shape)
    finally:
        blgnn.make_flags = orig
    assert preds.shape[0] == pairs


def test_expand_rows_moves_bytes(device):
    """blgnn_expand_rows: out[i] = src[row_map[i]] bit for bit -- fp32 rows, a column slice of fp16 piece planes
    (16-byte units, strided rows), 3 of 8 scale bytes per row (byte units), repeated and unordered source rows"""
    g = torch.Generator().manual_seed(11)
    nu, n = 333, 2100
    row_map = torch.randint(0, nu, (n,), generator=g).to(torch.int32).to(device)
    idx = row_map.long()
    x = torch.randn(nu, 256, generator=g).to(device)
    assert torch.equal(ops.expand_rows(x, row_map), x.index_select(0, idx))
    src, dst = ops.SpRows(nu, 1024, device), ops.SpRows(n, 1024, device)
    src.planes.copy_(torch.randn(nu, 2048, generator=g).to(device).half())
    src.scales.copy_(torch.randint(0, 255, (nu, 4), generator=g).to(torch.uint8).to(device))
    dst.planes.zero_()
    dst.scales.fill_(7)
    ops.expand_rows_sp(src.cols(0, 768), row_map, dst.cols(0, 768))
    assert torch.equal(dst.planes[:, :1536], src.planes[:, :1536].index_select(0, idx))
    assert torch.equal(dst.scales[:, :3], src.scales[:, :3].index_select(0, idx))
    assert float(dst.planes[:, 1536:].abs().max()) == 0.0 and bool((dst.scales[:, 3] == 7).all())   # the u block: untouched
    odd = torch.randn(nu, 13, generator=g).to(device)                 # 52-byte rows: the byte path
    assert torch.equal(ops.expand_rows(odd, row_map), odd.index_select(0, idx))
    with pytest.raises(ValueError):
        ops.expand_rows(x, row_map, out=torch.empty(n, 128, device=device))


@pytest.mark.parametrize("d", [256, 512])
def test_gmn_shared_prefix_equals_the_replicated_computation(device, d):
    """pair batch with every molecule in ~7 pairs: the first GIN layer and the first propagator's message half run on
    the distinct molecules (FLAGS.gmn_shared_prefix, default) -- every layer output must equal the forward that
    computes each copy (the reference's batch: src/batch.py:105-144 with batch_unique_graphs=False): the GIN layer
    bit for bit at d=256, everything behind it to fp32 rounding (K-split sum of f_node's first Linear)"""
    from helpers import build
    from test_gpu_configs import _pairs
    ds = blgnn.make_synthetic(90, 1200, shape="drugcombo", seed=5, device=device, num_edge_types=2)
    flags = blgnn.make_flags("lower_level_gmn", dataset="drugcombo", lower_level_gnn_type="gin", D_lower=d,
                             device=device, gmn_sp="force")
    model, data = build(flags, ds, seed=6, num_hyper_edge_feat=3)
    bd = blgnn.BatchData(_pairs(ds, 320), ds, is_train=False, unique_graphs=False)
    sm = bd.shared_molecules()
    n = bd.merge_data['merge'].x.shape[0]
    assert sm is not None and sm.graph.num_nodes < 0.3 * n
    assert torch.equal(sm.graph.x.index_select(0, sm.row_map.long()), bd.merge_data['merge'].x)

    from blgnn.config import FLAGS

    def run(shared):
        FLAGS.gmn_shared_prefix = shared
        model.acts = [bd.merge_data['merge'].x]
        for layer in model.layers[:-1]:
            model.acts.append(layer(model.acts[-1], bd, model))
        torch.cuda.synchronize()
        return list(model.acts[1:])
    a = run(True)
    assert getattr(a[0], "_blgnn_shared", None) is not None        # the path under test was taken
    b = run(False)
    assert getattr(b[0], "_blgnn_shared", None) is None
    for k, (u, v) in enumerate(zip(a, b)):
        if d == 256 and k == 0:
            # the fused GIN layer: a row's result does not depend on the other rows of the launch
            assert torch.equal(u, v), "layer %d differs" % k
        else:
            # from the first propagator on: f_node's first Linear is summed as [x | S] @ W[:3D] + u @ W[3D:] instead of
            # one chain over 4D columns (and at d=512 the unfused GIN layer's GEMMs pick their tile shape by the number
            # of rows): equal to fp32 rounding, not bit for bit
            assert max_err(u, v.cpu()) < 0.1 * TOL, "layer %d differs" % k


def test_gmn_pooled_tail_equals_the_readout_of_node_rows(device):
    """last propagator + multi-scale mean readout: hidden rows pooled in the GEMM epilogue and the second Linear applied per
    graph (FLAGS.gmn_pooled_tail, default) == the readout of the node rows, to fp32 rounding; the layer's output is
    then [G, D] tagged ``_blgnn_pooled``; pair scores agree"""
    from helpers import build
    from test_gpu_configs import _pairs
    from blgnn.config import FLAGS
    ds = blgnn.make_synthetic(90, 1200, shape="drugcombo", seed=8, device=device, num_edge_types=2)
    flags = blgnn.make_flags("lower_level_gmn", dataset="drugcombo", lower_level_gnn_type="gin", D_lower=256,
                             device=device, gmn_sp="force")
    model, data = build(flags, ds, seed=9, num_hyper_edge_feat=3)
    bd = blgnn.BatchData(_pairs(ds, 300), ds, is_train=False, unique_graphs=False)

    def run(tail):
        FLAGS.gmn_pooled_tail = tail
        model.acts = [bd.merge_data['merge'].x]
        for layer in model.layers[:-1]:
            model.acts.append(layer(model.acts[-1], bd, model))
        torch.cuda.synchronize()
        return list(model.acts[1:])
    a, b = run(True), run(False)
    G, n = bd.merge_data['merge'].num_graphs, bd.merge_data['merge'].x.shape[0]
    assert getattr(a[5], "_blgnn_pooled", None) == "mean" and tuple(a[5].shape) == (G, 256)
    for k in (1, 3):                                                   # earlier propagators: node rows + their pooled rows
        kind, pooled = a[k]._blgnn_pool_of
        assert kind == "mean" and tuple(pooled.shape) == (G, 256) and tuple(a[k].shape) == (n, 256)
        assert getattr(b[k], "_blgnn_pool_of", None) is None
        assert max_err(a[k], b[k].cpu()) < 0.1 * TOL
    assert getattr(b[5], "_blgnn_pooled", None) is None and tuple(b[5].shape) == (n, 256)
    assert tuple(a[6].shape) == tuple(b[6].shape) == (G, 768)
    assert max_err(a[6], b[6].cpu()) < 0.1 * TOL                       # the readout [G, 3 D]
    assert max_err(a[-1], b[-1].cpu()) < 0.1 * TOL                     # pair scores


@pytest.mark.parametrize("d", [256, 512])
def test_spmm_sum_sp_and_gin_as_presplit_gemms(device, d):
    """blgnn_spmm_sum_sp: (1 + eps) x_i + sum_j x_j straight into the sp format == the fp32 aggregation to 22 bits per
    256-column block; then the whole GIN layer as aggregation + two pre-split GEMMs with eval-BN, the readout on the
    second GEMM's epilogue, against fp64 (model/layers.py:9-63)"""
    g = torch.Generator().manual_seed(40 + d)
    n = 3000
    x = torch.randn(n, d, generator=g)
    src = torch.randint(0, n, (2 * n,), generator=g)
    dst = (src + torch.randint(1, 40, (2 * n,), generator=g)) % n
    ei = torch.stack([torch.cat([src, dst]), torch.cat([dst, src])])
    ei = torch.unique(ei, dim=1)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    ref_agg = 1.3 * x.double()
    ref_agg.index_add_(0, ei[0], x.double()[ei[1]])
    agg_sp = ops.spmm_sum_sp(csr, dv(x), self_coef=1.3, skip_self=True)
    assert _row_rel(ops.join_rows_sp(agg_sp), ref_agg) < 2.0 ** -20
    w1, w2 = torch.randn(d, d, generator=g) / d ** 0.5, torch.randn(d, d, generator=g) / d ** 0.5
    b1, b2 = torch.randn(d, generator=g) * 0.1, torch.randn(d, generator=g) * 0.1
    scale, shift = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g) * 0.1
    h = torch.relu(ref_agg @ w1.double() + b1.double())
    ref = torch.relu(h @ w2.double() + b2.double()) * scale.double() + shift.double()
    sizes = [30] * (n // 30)
    seg = torch.tensor([0] + sizes).cumsum(0).to(torch.int32)
    rg = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).to(torch.int32)
    pooled = torch.zeros(len(sizes), d, device=device)
    h_sp = ops.SpRows(n, d, device)
    ops.linear_sp(agg_sp, dv(w1), dv(b1), act="relu", out_sp=h_sp)
    y, _ = ops.linear_sp(h_sp, dv(w2), dv(b2), act="relu", bn_scale=dv(scale), bn_shift=dv(shift), f32=True,
                         pool=(dv(rg), dv(seg), pooled, "mean"))
    assert max_err(y, ref) < TOL
    assert max_err(pooled, ref.view(len(sizes), 30, d).mean(1)) < TOL
    with pytest.raises((ValueError, NotImplementedError)):
        ops.spmm_sum_sp(csr, dv(x)[:, :192].contiguous())


def test_gmn_sp_pipeline_without_side_stream(device):
    """the same forced sp pipeline with the attention on the launch stream (FLAGS.gmn_overlap_match = False): shared
    prefix, K-split, pooled readout and the in-line attention + split against the fp64 oracle"""
    from test_gpu_configs import _gmn_model_vs_oracle
    import blgnn.config as cfg
    orig = cfg.make_flags

    def forced(*a, **k):
        k.setdefault("gmn_sp", "force")
        f = orig(*a, **k)
        f.gmn_overlap_match = False
        return f

    blgnn.make_flags = forced
    try:
        preds = _gmn_model_vs_oracle(device, 256, 100, 1200, 200, 77, "drugcombo", 2, "drugcombo")
    finally:
        blgnn.make_flags = orig
    assert preds.shape[0] == 200


def test_gmn_match_sp_rejects_other_widths_and_aliased_planes(device):
    """one scale block per row is what the kernel is written for: d = 512 planes are refused (callers keep the fp32
    kernels), so are result planes aliasing the input's"""
    sizes = [10, 12, 20, 9]
    off = np.concatenate([[0], np.cumsum(sizes)])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    order = ops.match_pair_order(ind, device)
    x512 = ops.split_rows_sp(torch.randn(int(off[-1]), 512, device=device))
    with pytest.raises(NotImplementedError):
        ops.gmn_match_sp(seg, x512, ops.SpRows(int(off[-1]), 512, device), order)
    x256 = ops.split_rows_sp(torch.randn(int(off[-1]), 256, device=device))
    with pytest.raises(ValueError):
        ops.gmn_match_sp(seg, x256, x256, order)


@pytest.mark.parametrize("fork,streams,match_sp", [("message", 2, True), ("edge", 1, True), ("split", 1, True),
                                                   ("split", 1, False)])
def test_gmn_sp_pipeline_attention_placements(device, fork, streams, match_sp):
    """where and how the attention of the forced sp pipeline runs: forked beside the message GEMM / the edge kernel / the
    rare size classes beside the GEMM and the common ones beside the edge kernel, on one or two side streams, on the sp
    planes (blgnn_gmn_match_sp + the fp32 kernels for pairs above 64 nodes) or entirely in fp32 + split -- every variant
    against the fp64 oracle"""
    from test_gpu_configs import _gmn_model_vs_oracle
    import blgnn.config as cfg
    orig = cfg.make_flags

    def forced(*a, **k):
        k.setdefault("gmn_sp", "force")
        f = orig(*a, **k)
        f.gmn_match_fork, f.gmn_match_streams, f.gmn_match_sp = fork, streams, match_sp
        return f

    blgnn.make_flags = forced
    try:
        preds = _gmn_model_vs_oracle(device, 256, 100, 1200, 200, 77, "drugcombo", 2, "drugcombo")
    finally:
        blgnn.make_flags = orig
    assert preds.shape[0] == 200


@pytest.mark.parametrize("sizes,mapped", [
    ([32, 32, 33, 31, 64, 64, 1, 64, 64, 1, 5, 7, 24, 24], False),     # tile edges of both kernels, one-node graphs
    ([40, 38, 2, 60, 17, 9, 28, 0, 0, 13, 8, 8], False),               # an empty graph on either side
    ([29, 31] * 40 + [50, 47] * 30 + [64, 3], True),                   # x kept on distinct rows behind a row map
    ([12, 30, 64, 64, 65, 20, 128, 3, 129, 40, 7, 200, 33, 33], False),  # pairs above 64 nodes: fp32 kernels + their split
])
def test_gmn_match_sp(device, sizes, mapped):
    """the attention on the sp planes of x with an sp result (model/layers_gmn.py:55-84) against the fp64 oracle on the
    values the planes hold; rows of very different magnitude inside a pair (the partner scales ride on exp(cos), the
    result row's scale is a bound); every row's pieces must be finite and its scale byte written"""
    d = 256
    off = np.concatenate([[0], np.cumsum(sizes)])
    n = int(off[-1])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    g = torch.Generator().manual_seed(len(sizes))
    x = torch.randn(n, d, generator=g)
    x[::3] *= 37.0
    x[1::7] *= 1e-3
    if n > 20:
        x[11] = 0.0
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    order = ops.match_pair_order(ind, device)
    if mapped:                                                         # the planes hold a permutation of the rows (+ unused ones)
        perm = torch.randperm(n + 50, generator=g)[:n]
        xt = torch.zeros(n + 50, d)
        xt[perm] = x
        x_sp = ops.split_rows_sp(xt.to(device))
        row_map = perm.to(torch.int32).to(device)
    else:
        x_sp, row_map = ops.split_rows_sp(x.to(device)), None
    xh = ops.join_rows_sp(x_sp).cpu()                                  # what the planes hold (22 bits of x)
    xin = xh[perm] if mapped else xh
    ref = oracle.gmn_match(xin.double(), ind)
    out = ops.SpRows(n, 4 * d, device)
    out.planes.fill_(float("nan"))
    out.scales.fill_(255)
    u_sp = out.cols(3 * d, 4 * d)
    ops.gmn_match_sp(seg, x_sp, u_sp, order, row_map=row_map, x=xin.to(device))
    assert bool(torch.isfinite(u_sp.planes.float()).all()) and int(u_sp.scales.max()) < 255
    assert bool(torch.isnan(out.planes[:, :6 * d].float()).all())      # nothing outside its column slice
    got = ops.join_rows_sp(u_sp)
    assert max_err(got, ref) < TOL
    # per pair, relative to the pair's largest input magnitude (the unit the result's scale is bounded by)
    for k in range(len(sizes) // 2):
        a, b = ind[2 * k][0], ind[2 * k + 1][1]
        if b > a:
            scale = float(xin[a:b].abs().max())
            assert float((got[a:b].cpu().double() - ref[a:b]).abs().max()) <= 2e-6 * max(scale, 1e-30), k
