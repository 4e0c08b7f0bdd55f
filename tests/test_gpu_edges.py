"""GPU parity on awkward shapes for the matrix-core kernels added late in round 1: tile-edge row counts,
single rows, widths that are not multiples of anything, empty graphs / batches.  Small inputs, fp32 bar."""

import numpy as np
import pytest
import torch

import blgnn
from blgnn import ops
import oracle
from helpers import TOL, max_err

pytestmark = pytest.mark.gpu


def _graph(n, avg_deg, seed):
    if n < 2 or avg_deg == 0:
        return torch.zeros((2, 0), dtype=torch.int64)
    rng = np.random.default_rng(seed)
    e = rng.integers(0, n, size=(max(int(n * avg_deg / 2), 1), 2))
    return torch.as_tensor(blgnn.graph.to_undirected_host(e, n))


@pytest.mark.parametrize("transform", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("n,K,deg", [(1, 4, 0), (31, 52, 2), (32, 128, 3), (33, 100, 2), (65, 8, 0), (1000, 128, 5)])
def test_split_gin_layers_on_tile_edges(device, transform, n, K, deg):
    N = 128
    ei = _graph(n, deg, n + K)
    g = torch.Generator().manual_seed(n * 131 + K)
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    D = lambda t: t.double()
    ref = oracle.apply_act(oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), "relu", eps=0.0), "relu", None)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    if transform == "f16x3":
        assert ops.gin_layer_h3_supported(csr, dv(x), K, N)
        w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
        got = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, act="relu")
    else:
        assert ops.gin_layer_x6_supported(csr, dv(x), K, N)
        w1p, w2p = ops.pack_weight_x6(dv(w1.t().contiguous())), ops.pack_weight_x6(dv(w2.t().contiguous()))
        got = ops.gin_layer_x6(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, act="relu")
    assert got.shape == (n, N) and max_err(got, ref) < TOL


def test_f16x3_zero_and_constant_rows(device):
    """all-zero rows (scale guard) and rows of one repeated value go through the normalisation unharmed"""
    N, K, n = 128, 64, 96
    ei = _graph(n, 2, 7)
    x = torch.randn(n, K)
    x[::4] = 0.0
    x[1::4] = 3.25
    w1, w2 = torch.randn(N, K) / 8, torch.randn(N, N) / 11
    D = lambda t: t.double()
    zb = torch.zeros(N, dtype=torch.float64)
    ref = oracle.gin_conv(D(x), ei, D(w1), zb, D(w2), zb, "identity", eps=0.0)
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    got = ops.gin_layer_h3(csr, dv(x), ops.pack_weight_h3(dv(w1.t().contiguous())), None,
                           ops.pack_weight_h3(dv(w2.t().contiguous())), None, N)
    assert bool(torch.isfinite(got).all()) and max_err(got, ref) < TOL
    zero_w = ops.pack_weight_h3(torch.zeros(K, N, device=device))             # all-zero weights: scale guard
    out = ops.gin_layer_h3(csr, dv(x), zero_w, None, ops.pack_weight_h3(dv(w2.t().contiguous())), None, N)
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (31, 50, 3), (33, 128, 30), (1326, 128, 640), (1326, 64, 641),
                                   (8191, 7, 12), (100, 128, 1000)])
@pytest.mark.parametrize("group", ["src", "dst"])
def test_gat_transform_shapes(device, M, N, K, group):
    ei = _graph(M, 6, M + N + K)
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    x = torch.randn(M, K, generator=g)
    W = torch.randn(K, N, generator=g) / K ** 0.5
    att = torch.randn(1, 1, 2 * N, generator=g) / N ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    ref = oracle.gat_conv(x.double(), ei, W.double(), att.double(), b.double(), group)
    csr = ops.csr_from_edge_index(ei.to(device), M)
    got = ops.gat_conv(csr, x.to(device), W.to(device), att.to(device), b.to(device), group=group)
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("B", [1, 31, 32, 33, 95])
@pytest.mark.parametrize("d,dims,final", [(128, [32, 4, 1], "sigmoid"), (64, [16, 2, 3], "identity"), (256, [64, 8, 1], "sigmoid"),
                                          (8, [2, 1], "sigmoid")])
def test_link_head_batch_edges(device, B, d, dims, final):
    g = torch.Generator().manual_seed(B * 7 + d)
    z = torch.randn(40, d, generator=g)
    z[3] = 0.0                                                   # zero row: eps clamp of the normalisation
    ids = torch.randint(0, 40, (B, 2), generator=g)
    ids[0, 0] = 3
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


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gmn_match_random_pairs(device, seed):
    rng = np.random.default_rng(seed)
    sizes = rng.integers(0, 70, size=24).tolist()
    sizes[5] = 0
    sizes[6] = 64
    sizes[7] = 65
    d = int(rng.choice([64, 128, 320]))
    off = np.concatenate([[0], np.cumsum(sizes)])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    x = torch.randn(int(off[-1]), d, generator=torch.Generator().manual_seed(seed))
    ref = oracle.gmn_match(x.double(), ind)
    seg = torch.as_tensor(off, dtype=torch.int32, device=device)
    got = ops.gmn_match(seg, x.to(device))
    assert max_err(got, ref) < TOL


@pytest.mark.parametrize("M,N,K", [(32769, 3, 4), (16385, 130, 12), (40000, 128, 1000), (33000, 257, 36)])
def test_linear_bf16x6_odd_widths(device, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(K, N, generator=g) / K ** 0.5
    xd, wd = x.to(device), w.to(device)
    if not ops.linear_x6_supported(xd, wd):
        pytest.skip("shape stays on the f32 kernels")
    got = ops.linear(xd, wd, transform="bf16x6")
    assert max_err(got, x.double() @ w.double()) < TOL


@pytest.mark.parametrize("n_mols,K,N", [(1, 4, 256), (3, 252, 256), (5, 36, 256), (9, 160, 256), (9, 224, 256), (2, 256, 256),
                                         (3, 68, 128)])
def test_gin_layer_f16x3_odd_shapes_and_unaligned_output(device, n_mols, K, N):
    """the d = 256 layer kernel (64-row tiles, two workgroups per CU) off its fast paths: fewer rows than one tile, K that is
    not a multiple of 64 (zero-padded steps of the weight stream: K = 4, 36, 160, 224, 252), and an output slice whose
    leading dimension / base address rule out 16-byte stores (per-element stores with the row predicate)"""
    ds = blgnn.make_synthetic(n_mols, 10, seed=5 + K, device="cpu")
    merged = ds.molecules.merge(ds.molecules.gids)["merge"]
    ei, n = merged.edge_index, merged.x.shape[0]
    g = torch.Generator().manual_seed(K + N)
    x = torch.randn(n, K, generator=g)
    w1, b1 = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    w2, b2 = torch.randn(N, N, generator=g) / N ** 0.5, torch.randn(N, generator=g) * 0.1
    D = lambda t: t.double()
    ref = oracle.apply_act(oracle.gin_conv(D(x), ei, D(w1), D(b1), D(w2), D(b2), "relu", eps=0.5), "relu")
    dv = lambda t: t.to(device)
    csr = ops.csr_from_edge_index(dv(ei), n)
    assert ops.gin_layer_h3_supported(csr, dv(x), K, N)
    w1p, w2p = ops.pack_weight_h3(dv(w1.t().contiguous())), ops.pack_weight_h3(dv(w2.t().contiguous()))
    got = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.5, act="relu")
    assert max_err(got, ref) < TOL
    wide = torch.full((n, N + 3), -3.0, device=device)
    odd = ops.gin_layer_h3(csr, dv(x), w1p, dv(b1), w2p, dv(b2), N, eps=0.5, act="relu", out=wide[:, 1:1 + N])
    assert torch.equal(odd, got)
    assert float((wide[:, :1] + 3.0).abs().max()) == 0.0 and float((wide[:, 1 + N:] + 3.0).abs().max()) == 0.0
