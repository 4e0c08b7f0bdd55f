"""GPU parity, op level: every C-ABI entry point against the CPU oracle on the same
seeded inputs (bit-exact for integer layouts, fp32 max|delta| < 1e-5 for features)."""

import numpy as np
import pytest
import torch

import blgnn
from blgnn import ops
import oracle
from helpers import TOL, max_err

pytestmark = pytest.mark.gpu


def rand_graph(n, avg_deg, seed, self_loops=False, isolated=0):
    rng = np.random.default_rng(seed)
    m = max(int(n * avg_deg / 2), 1)
    e = rng.integers(0, n - isolated, size=(m, 2))
    if not self_loops:
        e = e[e[:, 0] != e[:, 1]]
    return torch.as_tensor(blgnn.graph.to_undirected_host(e, n))


def molecule_graph(num_mols, seed):
    ds = blgnn.make_synthetic(num_mols, 10, seed=seed, device="cpu")
    merged = ds.molecules.merge(ds.molecules.gids)
    return merged["merge"].edge_index, merged["merge"].batch, merged["merge"].x.shape[0]


def np_csr(key, nbr, n):
    order = np.lexsort((nbr, key))
    key, nbr = key[order], nbr[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, key + 1, 1)
    return np.cumsum(rowptr), nbr


@pytest.mark.parametrize("n,deg,seed", [(1, 0, 0), (7, 2, 1), (300, 5, 2), (5000, 40, 3), (70000, 3, 4)])
def test_csr_sorted_symmetric(device, n, deg, seed):
    ei = rand_graph(n, deg, seed, self_loops=True) if deg else torch.zeros((2, 0), dtype=torch.int64)
    csr = ops.csr_from_edge_index(ei.to(device), n, symmetric=True)
    rp, col = np_csr(ei[0].numpy(), ei[1].numpy(), n)
    assert np.array_equal(csr.rowptr.cpu().numpy(), rp)
    assert np.array_equal(csr.col.cpu().numpy(), col)


@pytest.mark.parametrize("n,m,seed", [(50, 400, 0), (2000, 30000, 1), (300, 40000, 2)])
def test_csr_unsorted_directed(device, n, m, seed):
    """arbitrary order, duplicates, non-symmetric, rows longer than one wavefront"""
    rng = np.random.default_rng(seed)
    ei = torch.as_tensor(rng.integers(0, n, size=(2, m)))
    csr = ops.csr_from_edge_index(ei.to(device), n, symmetric=False)
    rp, col = np_csr(ei[1].numpy(), ei[0].numpy(), n)
    assert np.array_equal(csr.rowptr.cpu().numpy(), rp)
    assert np.array_equal(csr.col.cpu().numpy(), col)


def test_segments_with_empty_graphs(device):
    batch = torch.as_tensor([0, 0, 2, 2, 2, 5], dtype=torch.int64)
    seg = ops.segments_from_batch(batch.to(device), 8).cpu().numpy()
    assert seg.tolist() == [0, 2, 2, 5, 5, 5, 6, 6, 6]
    seg = ops.segments_from_batch(torch.zeros(0, dtype=torch.int64, device=device), 3).cpu().numpy()
    assert seg.tolist() == [0, 0, 0, 0]


@pytest.mark.parametrize("d", [52, 40, 64, 128, 256, 320, 512, 1280, 2048, 50, 7])
def test_spmm_gin_aggregate(device, d):
    ei, _, n = molecule_graph(40, 5)
    x = torch.randn(n, d)
    ref = oracle.scatter_add(x[ei[0]], ei[1], n) + 1.0 * x
    csr = ops.csr_from_edge_index(ei.to(device), n)
    got = ops.spmm_sum(csr, x.to(device), self_coef=1.0, skip_self=True)
    assert max_err(got, ref) < TOL


def test_graph_without_edges(device):
    """batch of single-atom molecules: nnz = 0 everywhere on the path"""
    n, d = 37, 64
    ei = torch.zeros((2, 0), dtype=torch.int64, device=device)
    csr = ops.csr_from_edge_index(ei, n)
    x = torch.randn(n, d)
    assert max_err(ops.spmm_sum(csr, x.to(device), 1.0, True), x) < TOL
    W, att, b = torch.randn(d, d) / 8, torch.randn(1, 1, 2 * d), torch.randn(d)
    ref = oracle.gat_conv(x.double(), ei.cpu(), W.double(), att.double(), b.double(), "src")
    assert max_err(ops.gat_conv(csr, x.to(device), W.to(device), att.to(device), b.to(device)), ref) < TOL
    ref = oracle.gcn_conv(x.double(), ei.cpu(), W.double(), b.double())
    assert max_err(ops.gcn_conv(csr, x.to(device), W.to(device), b.to(device)), ref) < TOL
    w2 = torch.randn(d, d) / 8
    ref = torch.relu(oracle.gin_conv(x.double(), ei.cpu(), W.t().double(), b.double(), w2.t().double(), b.double(), "relu"))
    got = ops.gin_layer(csr, x.to(device), W.to(device), b.to(device), w2.to(device), b.to(device), act="relu")
    assert max_err(got, ref) < TOL


def test_spmm_weighted_selfloops_highdeg(device):
    n, d = 700, 128
    ei = rand_graph(n, 60, 7, self_loops=True, isolated=5)
    w = torch.rand(ei.shape[1])
    x = torch.randn(n, d)
    csr = ops.csr_from_edge_index(ei.to(device), n)
    ref = oracle.scatter_add(x[ei[1]] * w.view(-1, 1), ei[0], n) + 0.5 * x   # keyed by ei[0]
    got = ops.spmm_sum(csr, x.to(device), self_coef=0.5, skip_self=False, w=w.to(device))
    assert max_err(got, ref) < TOL
    # into a column slice of a wider buffer
    wide = torch.zeros(n, 3 * d, device=device)
    ops.spmm_sum(csr, x.to(device), self_coef=0.5, skip_self=False, w=w.to(device), out=wide[:, d:2 * d])
    assert max_err(wide[:, d:2 * d], ref) < TOL and float(wide[:, :d].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 3, 2), (500, 16, 128), (1000, 64, 52), (4096, 128, 128),
                                   (777, 256, 320), (300, 1, 16), (2048, 2048, 512), (50000, 64, 64),
                                   (129, 130, 131), (40000, 32, 40)])
def test_linear_shapes(device, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / max(K, 1) ** 0.5
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double() + b.double()
    got = ops.linear(x.to(device), w.to(device), b.to(device))
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("act", ["identity", "relu", "prelu", "sigmoid", "tanh"])
def test_linear_epilogue(device, act):
    M, N, K = 333, 96, 72
    x, w, b = torch.randn(M, K), torch.randn(K, N) / 8, torch.randn(N)
    slope, scale, shift = torch.rand(N), torch.rand(N) + 0.5, torch.randn(N)
    rv, rw = torch.rand(M) * 3, torch.randn(N)
    ref = x.double() @ w.double() + b.double() + rv.double().view(-1, 1) * rw.double().view(1, -1)
    ref = oracle.apply_act(ref, act, slope.double()) * scale.double() + shift.double()
    dv = lambda t: t.to(device)
    got = ops.linear(dv(x), dv(w), dv(b), act=act, act_param=dv(slope), bn_scale=dv(scale),
                     bn_shift=dv(shift), rowvec=dv(rv), rowvec_w=dv(rw))
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("K,act,N", [(52, "relu", 128), (128, "relu", 128), (128, "identity", 128), (64, "prelu", 128),
                                     (40, "tanh", 128), (128, "sigmoid", 128), (52, "relu", 256), (256, "relu", 256),
                                     (256, "identity", 256), (128, "prelu", 256), (100, "tanh", 256)])
def test_gin_layer_f16x3(device, K, act, N):
    """fp16 two-piece split GIN layer (row / matrix power-of-two scaling; N = 128: 32-row tiles, N = 256: 64-row
    tiles of a 512-thread workgroup) == oracle at the fp32 bar"""
    ei, _, n = molecule_graph(53, 21)                       # n is not a multiple of the 64-row tile
    g = torch.Generator().manual_seed(K * 7 + len(act))
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    slope, scale, shift = torch.rand(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = lambda t: t.double()
    ref = oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), act, eps=0.25, prelu_weight=D(slope))
    ref = oracle.apply_act(ref, act, D(slope)) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    assert ops.gin_layer_h3_supported(csr, dv(x), K, N)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    wide = torch.zeros(n, N + 8, device=device)
    amax = torch.full((n,), -1.0, device=device)
    got = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.25, act=act, act_param=dv(slope),
                           bn_scale=dv(scale), bn_shift=dv(shift), out=wide[:, 4:4 + N], amax_out=amax)
    assert max_err(got, ref) < TOL
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + N:].abs().max()) == 0.0
    assert torch.equal(amax, got.abs().amax(dim=1))          # row magnitudes out of the store loop


@pytest.mark.parametrize("N,kind,sizes", [(128, "mean", None), (128, "sum", None), (256, "mean", None),
                                          (128, "mean", [1, 70, 3, 129, 33, 2, 64])])
def test_gin_layer_f16x3_fused_readout(device, N, kind, sizes):
    """north_star "with the node readout fused in": the layer's epilogue adds every tile's per-graph column sums
    into a column slice of the pooled buffer; == scatter_mean / scatter_add of the layer's own output (fp32
    rounding apart: the mean weight is applied per tile) and == the oracle's readout of the oracle's layer.
    ``sizes``: graphs of 1 row, graphs spanning three and five tiles"""
    if sizes is None:
        ei, batch, n = molecule_graph(57, 21)
    else:
        rng = np.random.default_rng(1)
        edges, batch, n = [], [], 0
        for gi, sz in enumerate(sizes):
            for a in range(sz - 1):
                edges += [(n + a, n + a + 1), (n + a + 1, n + a)]
            batch += [gi] * sz
            n += sz
        ei = torch.as_tensor(sorted(edges), dtype=torch.int64).t().contiguous()
        batch = torch.as_tensor(batch, dtype=torch.int64)
    G = int(batch.max()) + 1
    K = N
    g = torch.Generator().manual_seed(N)
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    D = lambda t: t.double()
    ref = oracle.apply_act(oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), "relu", eps=0.0), "relu")
    ref_pool = oracle.node_aggregation([ref], batch, G, "avg_pool" if kind == "mean" else "sum")
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    seg = ops.segments_from_batch(dv(batch), G)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    pooled = torch.zeros(G, 3 * N, device=device)
    got = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, act="relu",
                           pool=(dv(batch).to(torch.int32), seg, pooled[:, N:2 * N], kind))
    plain = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, act="relu")
    assert torch.equal(got, plain)                                  # the layer's own output is untouched
    assert float(pooled[:, :N].abs().max()) == 0.0 and float(pooled[:, 2 * N:].abs().max()) == 0.0
    assert max_err(pooled[:, N:2 * N], ref_pool) < TOL
    assert max_err(pooled[:, N:2 * N], ops.segment_reduce(seg, got, kind)) < 1e-6


@pytest.mark.parametrize("act", ["relu", "identity"])
def test_gin_layer_f16x3_categorical_input(device, act):
    """first lower layer fed with 8 bytes of attribute value indices per atom == the same layer fed with the
    one-hot rows those indices stand for (bit for bit: same arithmetic) == oracle"""
    from blgnn.dataset import FEATURE_CARDS, one_hot_from_codes
    N = 128
    ei, _, n = molecule_graph(41, 21)
    rng = np.random.default_rng(7)
    codes6 = np.stack([rng.integers(0, c, size=n) for c in FEATURE_CARDS], axis=1)
    x = one_hot_from_codes(codes6, device="cpu")
    K = x.shape[1]
    codes = np.zeros((n, 8), dtype=np.int8)
    codes[:, :6] = codes6
    offsets = [0] + np.cumsum(FEATURE_CARDS).tolist()
    g = torch.Generator().manual_seed(3)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    scale, shift = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = lambda t: t.double()
    ref = oracle.apply_act(oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), act, eps=0.1), act) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    cd = torch.as_tensor(codes).to(device)
    assert ops.gin_layer_h3_codes_supported(csr, cd, offsets, N)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    a = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.1, act=act, bn_scale=dv(scale), bn_shift=dv(shift))
    b = ops.gin_layer_h3_codes(csr, cd, offsets, w1p, dv(b1), w2p, dv(b2), N, eps=0.1, act=act, bn_scale=dv(scale),
                               bn_shift=dv(shift))
    assert torch.equal(a, b)
    assert max_err(b, ref) < TOL


@pytest.mark.parametrize("xs,ws", [(1e-6, 1.0), (3e4, 1.0), (1.0, 1e-5), (1.0, 300.0), (1e5, 1e-4)])
def test_gin_layer_f16x3_range(device, xs, ws):
    """the power-of-two normalisation keeps the fp16 pieces in range for tiny / huge activations and weights,
    and for rows whose entries span many orders of magnitude"""
    N, K = 128, 128
    ei, _, n = molecule_graph(9, 21)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, K, generator=g) * xs
    x[::3] *= torch.logspace(-6, 0, K).view(1, -1)          # wide dynamic range inside a row
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5 * ws, torch.zeros(N)
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5 / ws, torch.zeros(N)
    D = lambda t: t.double()
    ref = oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), "relu", eps=0.0)
    ref = oracle.apply_act(ref, "relu", None)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    got = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, act="relu")
    assert bool(torch.isfinite(got).all())
    assert float((got.cpu().double() - ref).abs().max() / ref.abs().max()) < TOL


@pytest.mark.parametrize("transform", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("M,N,K", [(32768, 128, 128), (33001, 64, 52), (32800, 100, 36), (11000, 300, 260),
                                   (32769, 1, 8), (16500, 256, 1024), (9000, 1024, 512),
                                   # N % 256 == 0: the 128 x 256 / K-step-32 variant of the f16x3 kernel -- ragged M, K not a
                                   # multiple of the step, a single K step, one and several column tiles
                                   (33001, 256, 36), (8200, 512, 100), (40000, 768, 32), (20000, 256, 260)])
def test_linear_split_shapes(device, M, N, K, transform):
    """split GEMMs on the 16-bit matrix cores (large-M path of ops.linear: fp16 two-piece / bf16 three-piece)
    == fp64 oracle at the fp32 bar"""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / max(K, 1) ** 0.5
    b = torch.randn(N, generator=g)
    ref = x.double() @ w.double() + b.double()
    xd, wd = x.to(device), w.to(device)
    assert ops.linear_x6_supported(xd, wd)
    got = ops.linear(xd, wd, b.to(device), transform=transform)
    assert max_err(got, ref) < TOL
    exact = ops.linear(xd, wd, b.to(device), transform="f32")
    assert max_err(exact, ref) < TOL
    # the packed image follows in-place weight updates
    wd.mul_(2.0)
    assert max_err(ops.linear(xd, wd, b.to(device), transform=transform), 2.0 * (x.double() @ w.double()) + b.double()) < TOL


@pytest.mark.parametrize("transform", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("act", ["identity", "relu", "prelu", "sigmoid", "tanh"])
def test_linear_split_epilogue(device, act, transform):
    M, N, K = 33321, 96, 72
    g = torch.Generator().manual_seed(77)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g) / 8, torch.randn(N, generator=g)
    slope, scale, shift = torch.rand(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    rv, rw = torch.rand(M, generator=g) * 3, torch.randn(N, generator=g)
    ref = x.double() @ w.double() + b.double() + rv.double().view(-1, 1) * rw.double().view(1, -1)
    ref = oracle.apply_act(ref, act, slope.double()) * scale.double() + shift.double()
    dv = lambda t: t.to(device)
    wide = torch.zeros(M, N + 8, device=device)
    got = ops.linear(dv(x), dv(w), dv(b), act=act, act_param=dv(slope), bn_scale=dv(scale),
                     bn_shift=dv(shift), rowvec=dv(rv), rowvec_w=dv(rw), out=wide[:, 4:4 + N], transform=transform)
    assert max_err(got, ref) < TOL
    assert float(wide[:, :4].abs().max()) == 0.0 and float(wide[:, 4 + N:].abs().max()) == 0.0


def test_linear_f16x3_row_magnitudes(device):
    """blgnn_row_absmax and the GEMM epilogue's fused row maxima (the normalisation input of the next
    f16x3 GEMM) are exactly the row-wise max |.| of what was read / stored, N not a tile multiple"""
    M, N, K = 33003, 200, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, K, generator=g).to(device)
    w = (torch.randn(K, N, generator=g) / 16).to(device)
    b = torch.randn(N, generator=g).to(device)
    assert torch.equal(ops.row_absmax(x), x.abs().amax(dim=1))
    wide = torch.randn(M, K + 12, generator=g).to(device)
    assert torch.equal(ops.row_absmax(wide[:, 4:4 + K]), wide[:, 4:4 + K].abs().amax(dim=1))
    assert ops.linear_writes_amax(x, w, "f16x3")
    am = torch.zeros(M, device=device)
    y = ops.linear(x, w, b, act="tanh", transform="f16x3", amax_out=am)
    assert torch.equal(am, y.abs().amax(dim=1))
    # and it is a valid row_amax for the next GEMM
    w2 = (torch.randn(N, 64, generator=g) / 8).to(device)
    z = ops.linear(y, w2, transform="f16x3", row_amax=am)
    assert max_err(z, y.double().cpu() @ w2.double().cpu()) < TOL


@pytest.mark.parametrize("xs,ws", [(1e-6, 1.0), (3e4, 1.0), (1.0, 1e-5), (1.0, 300.0), (1e5, 1e-4)])
def test_linear_f16x3_range(device, xs, ws):
    """per-row / per-matrix power-of-two normalisation of the f16x3 GEMM: tiny and huge activations and
    weights, rows of very different magnitude in one panel, entries spanning six decades inside a row,
    an all-zero row"""
    M, N, K = 33000, 192, 256
    g = torch.Generator().manual_seed(11)
    x = torch.randn(M, K, generator=g) * xs
    x[::3] *= torch.logspace(-6, 0, K).view(1, -1)
    x[1::7] *= 1e-4                                          # whole rows far below their neighbours
    x[5] = 0.0
    w = torch.randn(K, N, generator=g) / K ** 0.5 * ws
    ref = x.double() @ w.double()
    got = ops.linear(x.to(device), w.to(device), transform="f16x3")
    assert bool(torch.isfinite(got).all())
    err = (got.cpu().double() - ref).abs()
    # row-wise bar: every row within 1e-5 of ITS OWN scale (rows differ by 1e4 and more)
    row_scale = ref.abs().amax(dim=1).clamp_min(1e-30)
    assert float((err.amax(dim=1) / row_scale)[ref.abs().amax(dim=1) > 0].max()) < TOL
    assert float(got[5].abs().max()) == 0.0


@pytest.mark.parametrize("K,N,act", [(52, 64, "relu"), (52, 128, "relu"), (40, 256, "relu"), (64, 64, "identity"),
                                     (128, 128, "relu"), (128, 128, "identity"), (256, 256, "relu"),
                                     (256, 128, "prelu"), (64, 256, "tanh")])
def test_gin_layer_fused(device, K, N, act):
    """single-launch GIN layer == oracle GINConv -> act -> eval BN (model/layers.py:51-57)"""
    ei, _, n = molecule_graph(37, 21)           # n is not a multiple of the 64-row tile
    g = torch.Generator().manual_seed(K * 1000 + N)
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    slope, scale, shift = torch.rand(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = lambda t: t.double()
    ref = oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), act, eps=0.25, prelu_weight=D(slope))
    ref = oracle.apply_act(ref, act, D(slope)) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    assert ops.gin_layer_supported(csr, dv(x), K, N)
    got = ops.gin_layer(csr, dv(x), dv(w1.t().contiguous()), dv(b1), dv(w2.t().contiguous()), dv(b2),
                        eps=0.25, act=act, act_param=dv(slope), bn_scale=dv(scale), bn_shift=dv(shift))
    assert max_err(got, ref) < TOL
    # same numbers as the unfused three-kernel path, into a column slice
    wide = torch.zeros(n, N + 8, device=device)
    ops.gin_layer(csr, dv(x), dv(w1.t().contiguous()), dv(b1), dv(w2.t().contiguous()), dv(b2),
                  eps=0.25, act=act, act_param=dv(slope), bn_scale=dv(scale), bn_shift=dv(shift),
                  out=wide[:, 4:4 + N])
    assert torch.equal(wide[:, 4:4 + N], got) and float(wide[:, :4].abs().max()) == 0.0


@pytest.mark.parametrize("K0,N,L", [(52, 128, 5), (52, 64, 3), (40, 256, 2), (128, 128, 1)])
def test_gin_stack_fused_lower_level(device, K0, N, L):
    """L GIN layers + multi-scale mean readout in one launch == layer-by-layer oracle"""
    ds = blgnn.make_synthetic(90, 10, seed=K0 + N, device="cpu")
    keep = [g for g in ds.molecules.gids if ds.molecules.sizes[ds.molecules.gid_to_pos[g]] <= 64]
    merged = ds.molecules.merge(keep)
    m = merged["merge"]
    n, G = m.x.shape[0], len(keep)
    sizes = np.asarray([b - a for a, b in merged["ind_list"]])
    g = torch.Generator().manual_seed(L)
    x0 = torch.randn(n, K0, generator=g)
    layers, x64, pooled_ref = [], x0.double(), []
    for l in range(L):
        K = K0 if l == 0 else N
        act = "relu" if l < L - 1 else "identity"
        w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
        w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
        scale, shift = torch.rand(N, generator=g) * 0.5 + 0.25, torch.randn(N, generator=g) * 0.1
        D = lambda t: t.double()
        x64 = oracle.apply_act(oracle.gin_conv(x64, m.edge_index, D(w1), D(b1), D(w2), D(b2), act), act) * D(scale) + D(shift)
        pooled_ref.append(oracle.scatter_mean(x64, m.batch, G))
        dv = lambda t: t.contiguous().to(device)
        layers.append({"w1": dv(w1.t()), "b1": dv(b1), "w2": dv(w2.t()), "b2": dv(b2), "act": act, "eps": 0.0,
                       "scale": dv(scale), "shift": dv(shift)})
    tiles = ops.molecule_tiles(sizes, 64)
    assert tiles is not None and int(np.diff(tiles[0]).max()) <= 64
    csr = ops.csr_from_edge_index(m.edge_index.to(device), n)
    seg = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=device)
    pooled = torch.zeros(G + 3, L * N + 8, device=device)
    rows = torch.arange(G, dtype=torch.int32, device=device).flip(0) + 3           # scattered output rows
    y_last = torch.empty(n, N, device=device)
    ops.gin_stack(csr, x0.to(device), torch.as_tensor(tiles[0], device=device), torch.as_tensor(tiles[1], device=device),
                  seg, layers, pooled[:, 4:4 + L * N], [l * N for l in range(L)], "mean", row_map=rows, y_last=y_last)
    ref = torch.cat(pooled_ref, dim=1)
    assert max_err(pooled[rows.long(), 4:4 + L * N], ref) < TOL
    assert max_err(y_last, x64) < TOL
    assert float(pooled[:3].abs().max()) == 0.0 and float(pooled[:, :4].abs().max()) == 0.0
    assert ops.molecule_tiles([10, 70, 3], 64) is None


@pytest.mark.parametrize("K0,N,L,act0", [(52, 128, 5, "relu"), (128, 128, 2, "prelu"), (24, 128, 3, "relu"), (100, 128, 1, "identity")])
def test_gin_stack_f16x3(device, K0, N, L, act0):
    """the f16x3 stack (blgnn_gin_stack_h3): L GIN layers with both GEMMs on the fp16 matrix cores + multi-scale
    readout in one launch == layer-by-layer fp64 oracle"""
    ds = blgnn.make_synthetic(90, 10, seed=K0 + N, device="cpu")
    keep = [g for g in ds.molecules.gids if ds.molecules.sizes[ds.molecules.gid_to_pos[g]] <= 64]
    merged = ds.molecules.merge(keep)
    m = merged["merge"]
    n, G = m.x.shape[0], len(keep)
    sizes = np.asarray([b - a for a, b in merged["ind_list"]])
    g = torch.Generator().manual_seed(L)
    x0 = torch.randn(n, K0, generator=g)
    layers, x64, pooled_ref = [], x0.double(), []
    for l in range(L):
        K = K0 if l == 0 else N
        act = act0 if l < L - 1 else "identity"
        slope = torch.rand(N, generator=g)
        w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
        w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
        scale, shift = torch.rand(N, generator=g) * 0.5 + 0.25, torch.randn(N, generator=g) * 0.1
        D = lambda t: t.double()
        x64 = oracle.apply_act(oracle.gin_conv(x64, m.edge_index, D(w1), D(b1), D(w2), D(b2), act, prelu_weight=D(slope)), act, D(slope)) * D(scale) + D(shift)
        pooled_ref.append(oracle.scatter_mean(x64, m.batch, G))
        dv = lambda t: t.contiguous().to(device)
        layers.append({"w1p": ops.pack_weight_h3(dv(w1.t())), "b1": dv(b1), "w2p": ops.pack_weight_h3(dv(w2.t())),
                       "b2": dv(b2), "N": N, "act": act, "eps": 0.0, "slope": dv(slope) if act == "prelu" else None,
                       "scale": dv(scale), "shift": dv(shift)})
    tiles = ops.molecule_tiles(sizes, 64)
    assert tiles is not None and int(np.diff(tiles[0]).max()) <= 64
    csr = ops.csr_from_edge_index(m.edge_index.to(device), n)
    seg = torch.as_tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=device)
    pooled = torch.zeros(G + 3, L * N + 8, device=device)
    rows = torch.arange(G, dtype=torch.int32, device=device).flip(0) + 3           # scattered output rows
    y_last = torch.empty(n, N, device=device)
    ops.gin_stack(csr, x0.to(device), torch.as_tensor(tiles[0], device=device), torch.as_tensor(tiles[1], device=device),
                  seg, layers, pooled[:, 4:4 + L * N], [l * N for l in range(L)], "mean", row_map=rows, y_last=y_last)
    ref = torch.cat(pooled_ref, dim=1)
    assert max_err(pooled[rows.long(), 4:4 + L * N], ref) < TOL
    assert max_err(y_last, x64) < TOL
    assert float(pooled[:3].abs().max()) == 0.0 and float(pooled[:, :4].abs().max()) == 0.0




@pytest.mark.parametrize("K,N,act", [(52, 128, "relu"), (128, 128, "relu"), (128, 128, "identity"), (40, 256, "relu"),
                                     (256, 256, "relu"), (256, 128, "prelu"), (64, 256, "tanh")])
def test_gin_layer_bf16x6(device, K, N, act):
    """bf16 three-piece split transform (six MFMAs per product) keeps the fp32 parity bar"""
    ei, _, n = molecule_graph(37, 23)
    g = torch.Generator().manual_seed(K * 7 + N)
    x = torch.randn(n, K, generator=g) * 3.0                       # not just O(1) inputs
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    slope, scale, shift = torch.rand(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    D = lambda t: t.double()
    ref = oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), act, eps=0.25, prelu_weight=D(slope))
    ref = oracle.apply_act(ref, act, D(slope)) * D(scale) + D(shift)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    assert ops.gin_layer_x6_supported(csr, dv(x), K, N)
    w1p, w2p = ops.pack_weight_x6(dv(w1.t().contiguous())), ops.pack_weight_x6(dv(w2.t().contiguous()))
    got = ops.gin_layer_x6(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.25, act=act, act_param=dv(slope),
                           bn_scale=dv(scale), bn_shift=dv(shift))
    exact = ops.gin_layer(csr, dv(x), dv(w1.t().contiguous()), dv(b1), dv(w2.t().contiguous()), dv(b2), eps=0.25,
                          act=act, act_param=dv(slope), bn_scale=dv(scale), bn_shift=dv(shift))
    e6, e32 = max_err(got, ref), max_err(exact, ref)
    assert e6 < TOL and e6 < 4 * e32 + 2e-7, (e6, e32)            # as accurate as the exact-f32 MFMA path


def test_gin_layer_unsupported_shapes_are_reported(device):
    ei, _, n = molecule_graph(5, 1)
    csr = ops.csr_from_edge_index(ei.to(device), n)
    x = torch.randn(n, 50, device=device)
    assert not ops.gin_layer_supported(csr, x, 50, 64)
    with pytest.raises(NotImplementedError):
        ops.gin_layer(csr, x, torch.randn(50, 96, device=device), None, torch.randn(96, 96, device=device), None)


@pytest.mark.parametrize("group", ["src", "dst"])
@pytest.mark.parametrize("din,d", [(52, 64), (320, 64), (128, 128), (30, 50)])
def test_gat_conv(device, group, din, d):
    n = 400
    ei = rand_graph(n, 30, 11, self_loops=True, isolated=3)
    x = torch.randn(n, din)
    W = torch.randn(din, d) / din ** 0.5
    att = torch.randn(1, 1, 2 * d) / d ** 0.5
    b = torch.randn(d) * 0.1
    ref = oracle.gat_conv(x.double(), ei, W.double(), att.double(), b.double(), group)
    csr = ops.csr_from_edge_index(ei.to(device), n)
    got = ops.gat_conv(csr, x.to(device), W.to(device), att.to(device), b.to(device), group=group)
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("d", [64, 128, 320])
def test_convs_on_large_high_degree_graph(device, d):
    """5000 rows with mean degree 24 and a heavy tail: the shape of an 8-GPU interaction level"""
    n = 5000
    ei = rand_graph(n, 24, 31, self_loops=True, isolated=7)
    x = torch.randn(n, d)
    W, att, b = torch.randn(d, d) / d ** 0.5, torch.randn(1, 1, 2 * d) / d ** 0.5, torch.randn(d) * 0.1
    csr = ops.csr_from_edge_index(ei.to(device), n)
    dv = lambda t: t.to(device)
    for group in ("src", "dst"):
        ref = oracle.gat_conv(x.double(), ei, W.double(), att.double(), b.double(), group)
        assert max_err(ops.gat_conv(csr, dv(x), dv(W), dv(att), dv(b), group=group), ref) < TOL
    ref = oracle.gcn_conv(x.double(), ei, W.double(), b.double())
    assert max_err(ops.gcn_conv(csr, dv(x), dv(W), dv(b)), ref) < TOL
    ref = oracle.scatter_add(x[ei[1]].double(), ei[0], n) + x.double()
    assert max_err(ops.spmm_sum(csr, dv(x), 1.0, False), ref) < TOL


@pytest.mark.parametrize("d", [64, 128, 256, 320])
def test_convs_one_wavefront_per_row_with_hub_rows(device, d):
    """>= 16 384 rows at mean degree 20: blgnn_gat_aggregate / gcn / spmm take the one-wavefront-per-row kernel
    (k_gather_wave) for rows up to 384 edges and the workgroup-per-row kernel for the hubs behind it; the graph has hub
    rows of 700 and 1 500 edges, rows of exactly 64 / 65 / 384 / 385 edges (the pass and hand-over boundaries), isolated
    rows and self loops; a row subset (the pruned last interaction layer) must give the same bits as the full pass"""
    n = 20000
    rng = np.random.default_rng(d)
    e = [rng.integers(0, n - 5, size=(n * 10, 2))]
    for hub, deg in ((17, 700), (4242, 1500), (100, 64), (101, 65), (102, 384), (103, 385)):
        nb = rng.choice(n - 5, size=deg, replace=False)
        e.append(np.stack([np.full(deg, hub), nb], axis=1))
    ei = torch.as_tensor(blgnn.graph.to_undirected_host(np.concatenate(e), n))
    x = torch.randn(n, d)
    W, att, b = torch.randn(d, d) / d ** 0.5, torch.randn(1, 1, 2 * d) / d ** 0.5, torch.randn(d) * 0.1
    csr = ops.csr_from_edge_index(ei.to(device), n)
    deg_of = (csr.rowptr[1:] - csr.rowptr[:-1]).cpu()
    assert int(deg_of.max()) > 1000 and int((deg_of > 384).sum()) >= 3
    dv = lambda t: t.to(device)
    for group in ("src", "dst"):
        ref = oracle.gat_conv(x.double(), ei, W.double(), att.double(), b.double(), group)
        got = ops.gat_conv(csr, dv(x), dv(W), dv(att), dv(b), group=group)
        assert max_err(got, ref) < TOL
        rows = torch.as_tensor(rng.choice(n, size=17000, replace=False), device=device)
        part = ops.gat_conv(csr, dv(x), dv(W), dv(att), dv(b), group=group, rows=rows,
                            out=torch.full((n, d), -7.0, device=device))
        assert torch.equal(part[rows], got[rows])
        untouched = torch.ones(n, dtype=torch.bool, device=device)
        untouched[rows] = False
        assert float((part[untouched] + 7.0).abs().max()) == 0.0
        # a shorter list (>= 8 192 listed rows of a graph this large still take the wavefront kernel) holding both hubs,
        # with negative padding entries: skipped by the wavefront role and by the hub scan
        some = torch.as_tensor(np.concatenate([[17, 4242, 103], rng.choice(n, size=9000, replace=False)]), device=device)
        padded = torch.where(torch.arange(some.numel(), device=device) % 5 == 3, torch.full_like(some, -1), some)
        part = ops.gat_conv(csr, dv(x), dv(W), dv(att), dv(b), group=group, rows=padded,
                            out=torch.full((n, d), -7.0, device=device))
        kept = padded[padded >= 0]
        assert torch.equal(part[kept], got[kept])
        untouched = torch.ones(n, dtype=torch.bool, device=device)
        untouched[kept] = False
        assert float((part[untouched] + 7.0).abs().max()) == 0.0
    ref = oracle.gcn_conv(x.double(), ei, W.double(), b.double())
    assert max_err(ops.gcn_conv(csr, dv(x), dv(W), dv(b)), ref) < TOL
    ref = oracle.scatter_add(x[ei[1]].double(), ei[0], n) + x.double()
    assert max_err(ops.spmm_sum(csr, dv(x), 1.0, False), ref) < TOL


def test_gat_conv_molecules_and_accumulate(device):
    ei, _, n = molecule_graph(30, 3)
    d = 64
    x = torch.randn(n, d)
    params = [(torch.randn(d, d) / 8, torch.randn(1, 1, 2 * d) / 8, torch.randn(d) * 0.1) for _ in range(2)]
    eis = [ei, rand_graph(n, 4, 5)]
    ref = torch.relu(oracle.multi_edge_gat(x.double(), eis, [tuple(t.double() for t in p) for p in params], "src"))
    out = None
    for i, (e, (W, a, b)) in enumerate(zip(eis, params)):
        csr = ops.csr_from_edge_index(e.to(device), n)
        out = ops.gat_conv(csr, x.to(device), W.to(device), a.to(device), b.to(device), group="src",
                           act="relu" if i == 1 else "identity", out=out, accumulate=i > 0)
    assert max_err(out, ref) < TOL


@pytest.mark.parametrize("weighted", [False, True])
def test_gcn_conv(device, weighted):
    n, din, d = 500, 52, 128
    ei = rand_graph(n, 12, 13, self_loops=True, isolated=4)
    x, W, b = torch.randn(n, din), torch.randn(din, d) / 7, torch.randn(d) * 0.1
    ew = torch.rand(ei.shape[1]) + 0.1 if weighted else None
    if weighted:   # symmetric weights so that CSR-by-row == CSR-by-col
        key = torch.minimum(ei[0], ei[1]) * n + torch.maximum(ei[0], ei[1])
        ew = (key.double() * 0.6180339887).frac().float() + 0.1
    ref = oracle.gcn_conv(x.double(), ei, W.double(), b.double(), None if ew is None else ew.double())
    csr = ops.csr_from_edge_index(ei.to(device), n)
    got = ops.gcn_conv(csr, x.to(device), W.to(device), b.to(device),
                       None if ew is None else ew.to(device), act="identity")
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("kind", ["mean", "sum"])
@pytest.mark.parametrize("d", [64, 256, 50, 1280])
def test_segment_reduce(device, kind, d):
    sizes = [3, 0, 1, 40, 457, 0, 2, 17]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.as_tensor(sizes))
    x = torch.randn(int(sum(sizes)), d)
    f = oracle.scatter_mean if kind == "mean" else oracle.scatter_add
    ref = f(x.double(), batch, len(sizes))
    seg = ops.segments_from_batch(batch.to(device), len(sizes))
    got = ops.segment_reduce(seg, x.to(device), kind)
    assert max_err(got, ref) < TOL
    # multi-scale style: column slice + row redirection
    wide = torch.zeros(20, 2 * d + 4, device=device)
    rows = torch.as_tensor([19, 3, 5, 7, 11, 13, 0, 2], dtype=torch.int32, device=device)
    ops.segment_reduce(seg, x.to(device), kind, out=wide[:, 4:4 + d], row_map=rows)
    assert max_err(wide[rows.long(), 4:4 + d], ref) < TOL


def test_segment_reduce_multi_scale(device):
    sizes = [3, 0, 1, 40, 200, 2]
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.as_tensor(sizes))
    n = int(sum(sizes))
    wide = torch.randn(n, 300)
    xs = [torch.randn(n, 64), torch.randn(n, 128), wide[:, 10:60], torch.randn(n, 52), torch.randn(n, 256)]
    ref = oracle.node_aggregation([x.double() for x in xs], batch, len(sizes), "avg_pool")
    seg = ops.segments_from_batch(batch.to(device), len(sizes))
    dw = wide.to(device)
    got = ops.segment_reduce_multi(seg, [xs[0].to(device), xs[1].to(device), dw[:, 10:60], xs[3].to(device),
                                         xs[4].to(device)], "mean")
    assert got.shape == ref.shape and max_err(got, ref) < TOL


@pytest.mark.parametrize("d,sizes", [(64, [5, 7, 1, 3, 24, 24]), (256, [40, 38, 2, 70, 64, 65]),
                                     (512, [30, 31]), (256, [150, 170]), (128, [457, 12]),
                                     (256, [32, 32, 33, 31, 64, 64, 1, 64, 64, 1]),     # tile edges of the matrix-core kernel
                                     (192, [28, 40, 0, 17, 9, 0]),                      # empty graphs on either side
                                     (72, [20, 30, 44, 12]),                            # d % 64 != 0: scalar kernel
                                     (128, [28] * 64),
                                     # 65 .. 128 nodes: the LDS-resident matrix-core kernel (tile edges, one side small,
                                     # several per workgroup stride, next to small and > 128 pairs)
                                     (256, [65, 3, 128, 128, 97, 64, 1, 96, 30, 30, 129, 20, 100, 33]),
                                     (64, [70, 90] * 9 + [10, 12])])
def test_gmn_match(device, d, sizes):
    off = np.concatenate([[0], np.cumsum(sizes)])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    x = torch.randn(int(off[-1]), d)
    ref = oracle.gmn_match(x.double(), ind)
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    floor = torch.rand(int(off[-1]), device=device) * 0.5           # what another producer already put there
    am = floor.clone()
    got = ops.gmn_match(seg, x.to(device), row_amax=am)
    assert max_err(got, ref) < TOL
    # row magnitudes merged by atomic maximum: exactly max(previous, max_c |u|), whichever kernel took the pair
    assert torch.equal(am, torch.maximum(floor, got.abs().amax(dim=1)))
    # without the row magnitudes the small pairs take the wavefront-per-pair kernels (another summation order)
    plain = ops.gmn_match(seg, x.to(device))
    assert max_err(plain, ref) < TOL and max_err(plain, got) < 1e-5


def test_gmn_match_size_classes_and_hint(device):
    """the three size-class kernels are independent: issued one class at a time into one buffer, or with the
    largest-graph hint, the result is bit for bit that of the single call"""
    sizes = [12, 30, 64, 64, 65, 20, 128, 3, 129, 40, 7, 200, 33, 33]   # classes 1, 2, 3, 3, 4, 4, 2
    off = np.concatenate([[0], np.cumsum(sizes)])
    d = 256
    x = torch.randn(int(off[-1]), d, generator=torch.Generator().manual_seed(9)).to(device)
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    ref = ops.gmn_match(seg, x)
    out = torch.full_like(ref, float("nan"))
    for cls in (4, 1, 2):
        ops.gmn_match(seg, x, out=out, classes=cls)
    assert torch.equal(out, ref)
    assert torch.equal(ops.gmn_match(seg, x, max_nodes=max(sizes)), ref)
    # the same pairs handed over as per-class lists (one wavefront per listed pair instead of scanning)
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    order = ops.match_pair_order(ind, device)
    assert order[1].tolist() == [1, 2, 2, 2] and sorted(order[0].tolist()) == list(range(7))
    assert torch.equal(ops.gmn_match(seg, x, order=order), ref)
    # a batch whose largest graph has 64 nodes needs only the first kernel
    small = [12, 30, 64, 64, 20, 5]
    off2 = np.concatenate([[0], np.cumsum(small)])
    seg2 = torch.as_tensor(off2, dtype=torch.int32, device=device)
    x2 = x[:int(off2[-1])].contiguous()
    assert torch.equal(ops.gmn_match(seg2, x2, max_nodes=64), ops.gmn_match(seg2, x2))


def test_gmn_match_on_operands_beyond_4_gib(device):
    """the output of the match is a column slice of the [Nn, 4D] f_node input; at configs[4] (657 k atoms, D = 512)
    that buffer is 5.4 GB, so byte offsets from its base do not fit 32 bits -- pairs at both ends and in the middle
    must still land in their own rows"""
    sizes = [20, 21] * 14000 + [70, 90, 33, 2]
    off = np.concatenate([[0], np.cumsum(sizes)])
    n, d, ld = int(off[-1]), 64, 2048
    assert n * ld * 4 > 4.6e9
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, d, generator=g)
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    wide = torch.zeros((n, ld), device=device)
    got = ops.gmn_match(seg, x.to(device), out=wide[:, 1024:1024 + d])
    for first in (0, 6998, 13996, 13998, 14000):                 # pair index -> its two graphs
        a, b = int(off[2 * first]), int(off[2 * first + 2])
        ind = [(0, int(off[2 * first + 1]) - a), (int(off[2 * first + 1]) - a, b - a)]
        ref = oracle.gmn_match(x[a:b].double(), ind)
        assert max_err(got[a:b], ref) < TOL
    assert float(wide[:, :1024].abs().max()) == 0.0 and float(wide[:, 1024 + d:].abs().max()) == 0.0
    del wide
    torch.cuda.empty_cache()


def test_edge_relu_sum_matches_message(device):
    """restructured GMN message == reference formulation (layers_gmn.py:38-42)"""
    ei, _, n = molecule_graph(25, 9)
    D = 64
    x = torch.randn(n, D)
    lin = [(torch.randn(2 * D, 2 * D) / 11, torch.randn(2 * D) * 0.1) for _ in range(2)]
    ref = oracle.gmn_message_sum(x.double(), ei, [(w.double(), b.double()) for w, b in lin])
    (w1, b1), (w2, b2) = lin
    wpq = torch.cat([w1[:, :D].t(), w1[:, D:].t()], dim=1).contiguous()
    bpq = torch.cat([b1, torch.zeros_like(b1)])
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    pq = ops.linear(dv(x), dv(wpq), dv(bpq))
    am = torch.full((n,), 0.25, device=device)
    h = ops.edge_relu_sum(csr, pq[:, :2 * D], pq[:, 2 * D:], row_amax=am)
    assert torch.equal(am, torch.clamp(h.amax(dim=1), min=0.25))      # merged row maxima (this width: fallback pass)
    wide_p, wide_q = torch.randn(n, 512, device=device), torch.randn(n, 512, device=device)
    am2 = torch.zeros(n, device=device)
    h2 = ops.edge_relu_sum(csr, wide_p, wide_q, row_amax=am2)          # d4 = 128: the kernel's own epilogue
    assert torch.equal(am2, h2.amax(dim=1)) and torch.equal(h2, ops.edge_relu_sum(csr, wide_p, wide_q))
    got = ops.linear(h, dv(w2.t().contiguous()), None, rowvec=ops.row_degree(csr), rowvec_w=dv(b2))
    assert max_err(got, ref) < TOL


def test_pair_gather_and_normalize(device):
    z = torch.randn(100, 192)
    z[7] = 0.0
    ids = torch.randint(0, 100, (257, 2))
    ids[0, 0] = 7
    zn = oracle.l2_normalize(z.double())
    ref = torch.cat([zn[ids[:, 0]], zn[ids[:, 1]]], dim=1)
    got = ops.pair_gather_normalize(z.to(device), ids[:, 0].contiguous().to(device), ids[:, 1].contiguous().to(device))
    assert max_err(got, ref) < TOL
    assert max_err(ops.l2_normalize_rows(z.to(device)), zn) < TOL


@pytest.mark.parametrize("d,dims,final", [(128, [32, 4, 1], "sigmoid"), (64, [16, 2, 3], "identity"),
                                          (320, [80, 10, 1], "sigmoid"), (50, [7], "identity"),
                                          (256, [64, 8, 1], "sigmoid"), (96, [24, 3, 1], "sigmoid"),
                                          (64, [3], "identity"), (200, [100, 50, 3], "identity")])
def test_link_pred_head_fused(device, d, dims, final):
    g = torch.Generator().manual_seed(d)
    z = torch.randn(300, d, generator=g)
    z[11] = 0.0
    ids = torch.randint(0, 300, (1000, 2), generator=g)
    ids[0, 1] = 11
    widths = [2 * d] + dims
    lins = [(torch.randn(widths[i + 1], widths[i], generator=g) / widths[i] ** 0.5,
             torch.randn(widths[i + 1], generator=g) * 0.1) for i in range(len(dims))]
    ref = oracle.link_pred(z.double(), ids[:, 0], ids[:, 1], [(w.double(), b.double()) for w, b in lins],
                           "mlp_concat", multi_label_pred=(final == "identity"))
    dv = lambda t: t.to(device)
    got = ops.link_pred_head(dv(z), dv(ids[:, 0].contiguous()), dv(ids[:, 1].contiguous()),
                             [dv(w.t().contiguous()) for w, _ in lins], [dv(b) for _, b in lins],
                             hidden_act="relu", final_act=final)
    assert got.shape == ref.shape and max_err(got, ref) < TOL


def test_link_pred_dot_product_module(device):
    """LinkPred(type='dot_product') (model/layers_link_pred.py, non-default): sigmoid(<normalised z1, normalised z2>)"""
    import types
    from blgnn.layers_link_pred import LinkPred
    g = torch.Generator().manual_seed(4)
    z = torch.randn(90, 96, generator=g)
    ids = torch.randint(0, 90, (300, 2), generator=g)
    ref = oracle.link_pred(z.double(), ids[:, 0], ids[:, 1], None, "dot_product")
    lp = LinkPred("dot_product", 96, 1)
    bd = types.SimpleNamespace(_pair_row_ids=(ids[:, 0].contiguous().to(device), ids[:, 1].contiguous().to(device)))
    got = lp(z.to(device), bd, None)
    assert max_err(got.view(-1), ref.view(-1)) < TOL


def test_errors_are_loud(device):
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(4, 4), torch.randn(4, 4))          # CPU tensors: no fallback
    x = torch.randn(8, 8, device=device)
    with pytest.raises(ValueError):
        ops.linear(x, torch.randn(7, 8, device=device))
    with pytest.raises(ValueError):
        ops.linear(x, x, act="gelu")
