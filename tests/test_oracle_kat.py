"""CPU: hand-computed known-answer tests that pin the oracle (the reference ships no
tests or golden vectors -- SURVEY section 4 -- so these are the oracle's anchor).
Graphs: path P3 (0-1-2), star, two disjoint triangles, 1- and 2-node graphs."""

import math

import pytest
import torch

import oracle

P3 = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
T = lambda *rows: torch.tensor(rows, dtype=torch.float64)


def test_to_undirected_sorts_dedups_and_keeps_self_loop():
    ei = torch.tensor([[2, 0, 0, 1], [0, 1, 1, 1]])
    out = oracle.to_undirected(ei, 3)
    assert out.tolist() == [[0, 0, 1, 1, 2], [1, 2, 0, 1, 0]]
    assert oracle.to_undirected(torch.zeros((2, 0), dtype=torch.long), 4).shape == (2, 0)


def test_coalesce_max_keeps_larger_attr():
    ei = torch.tensor([[0, 1], [1, 0]])
    attr = T([1.0, 5.0], [3.0, 2.0])
    out, red = oracle.coalesce_max(ei, attr, 2)
    assert out.tolist() == [[0, 1], [1, 0]]
    assert red.tolist() == [[3.0, 5.0], [3.0, 5.0]]


def test_gin_path_graph():
    x = T([1.0], [2.0], [4.0])
    out = oracle.gin_conv(x, P3, T([1.0]), T(0.0).view(1), T([2.0]), T(1.0).view(1), "relu")
    # (1+0)x + sum_nbrs = [3, 7, 6]; nn = 2*relu(.)+1
    assert out.view(-1).tolist() == [7.0, 15.0, 13.0]
    # self loops are removed, eps scales only the self term
    loops = torch.cat([P3, torch.tensor([[1], [1]])], dim=1)
    out = oracle.gin_conv(x, loops, T([1.0]), T(0.0).view(1), T([1.0]), T(0.0).view(1), "identity", eps=0.5)
    assert out.view(-1).tolist() == [1.5 + 2, 3.0 + 5, 6.0 + 2]


def test_gin_shared_activation_is_identity_for_last_layer():
    x = T([-1.0], [2.0], [-4.0])
    out = oracle.gin_conv(x, P3, T([1.0]), T(0.0).view(1), T([1.0]), T(0.0).view(1), "identity")
    assert out.view(-1).tolist() == [1.0, -3.0, -2.0]        # no relu anywhere


def test_gcn_path_graph():
    x = T([1.0], [2.0], [4.0])
    out = oracle.gcn_conv(x, P3, T([1.0]), T(0.0).view(1))
    s6 = 1 / math.sqrt(6)
    exp = [1 / 2 + 2 * s6, 1 * s6 + 2 / 3 + 4 * s6, 2 * s6 + 4 / 2]
    assert out.view(-1).tolist() == pytest.approx(exp, abs=1e-12)
    # an existing self loop is replaced by the unit one (remove + add)
    loops = torch.cat([P3, torch.tensor([[0], [0]])], dim=1)
    assert oracle.gcn_conv(x, loops, T([1.0]), T(0.0).view(1)).view(-1).tolist() == pytest.approx(exp, abs=1e-12)
    # isolated node: deg = 1 (its self loop) -> out = x
    iso = oracle.gcn_conv(T([3.0], [5.0]), torch.zeros((2, 0), dtype=torch.long), T([1.0]), T(0.5).view(1))
    assert iso.view(-1).tolist() == [3.5, 5.5]


def test_gat_uniform_attention_both_groupings():
    h = T([1.0, 0.0], [0.0, 2.0], [4.0, 4.0])
    W, att, b = torch.eye(2, dtype=torch.float64), torch.zeros(1, 1, 4, dtype=torch.float64), T(0.0, 0.0).view(-1)
    dst = oracle.gat_conv(h, P3, W, att, b, "dst")
    assert dst.tolist() == [[0.5, 1.0], pytest.approx([5 / 3, 2.0]), [2.0, 3.0]]
    src = oracle.gat_conv(h, P3, W, att, b, "src")
    # alpha(j->i) = 1/(deg_j + 1): deg = [1, 2, 1]
    exp = [[1 / 2 + 0, 0 + 2 / 3], [1 / 2 + 0 + 4 / 2, 2 / 3 + 4 / 2], [4 / 2, 2 / 3 + 4 / 2]]
    for got, e in zip(src.tolist(), exp):
        assert got == pytest.approx(e, abs=1e-12)


def test_gat_two_nodes_by_hand():
    # h0 = 1, h1 = 3 (d=1); att = [a_dst, a_src] = [1, -1]; slope 0.2
    h = T([1.0], [3.0])
    ei = torch.tensor([[0, 1], [1, 0]])
    W, att, b = T([1.0]), T(1.0, -1.0).view(1, 1, 2), T(0.25).view(1)
    lr = lambda v: v if v > 0 else 0.2 * v
    s = lambda t, j: lr(1.0 * h[t].item() - 1.0 * h[j].item())      # target t, source j
    # dst grouping, node 0: edges (1->0), (0->0)
    e10, e00, e01, e11 = math.exp(s(0, 1)), math.exp(s(0, 0)), math.exp(s(1, 0)), math.exp(s(1, 1))
    out = oracle.gat_conv(h, ei, W, att, b, "dst").view(-1).tolist()
    assert out == pytest.approx([(e10 * 3 + e00 * 1) / (e10 + e00) + 0.25,
                                 (e01 * 1 + e11 * 3) / (e01 + e11) + 0.25], abs=1e-12)
    # src grouping: normalise over each SOURCE's out-edges: source 1 -> {0, 1}, source 0 -> {1, 0}
    out = oracle.gat_conv(h, ei, W, att, b, "src").view(-1).tolist()
    assert out == pytest.approx([e10 / (e10 + e11) * 3 + e00 / (e00 + e01) * 1 + 0.25,
                                 e01 / (e01 + e00) * 1 + e11 / (e11 + e10) * 3 + 0.25], abs=1e-12)


def test_gat_three_node_path_asymmetric_attention_by_hand():
    """3-node path 0-1-2, d = 2, att[:d] != att[d:]: every alpha written out edge by edge from the 1.1.x
    ``message()`` formula quoted in SURVEY section 8c(5) -- alpha_e = leaky_relu(h_i . att[:d] + h_j . att[d:]) for the
    edge j -> i, softmax over the edges sharing edge_index[0] ('src': the edges LEAVING a node) or edge_index[1]
    ('dst': the edges ENTERING it).  The two groupings give different numbers here, so the test pins which is which."""
    h = [[1.0, 2.0], [0.5, -1.0], [-2.0, 0.25]]
    a_dst, a_src = [0.7, -0.3], [-0.4, 0.9]
    edges = [(1, 0), (0, 1), (2, 1), (1, 2), (0, 0), (1, 1), (2, 2)]          # (source j, target i), self loops appended
    lr = lambda v: v if v > 0 else 0.2 * v
    dot = lambda u, v: sum(p * q for p, q in zip(u, v))
    score = {(j, i): lr(dot(h[i], a_dst) + dot(h[j], a_src)) for j, i in edges}
    exp = {}
    for group in ("src", "dst"):
        key = (lambda j, i: j) if group == "src" else (lambda j, i: i)
        den = {}
        for (j, i), sc in score.items():
            den[key(j, i)] = den.get(key(j, i), 0.0) + math.exp(sc)
        out = [[0.1, -0.2] for _ in range(3)]                                # bias
        for (j, i), sc in score.items():
            al = math.exp(sc) / den[key(j, i)]
            out[i] = [o + al * v for o, v in zip(out[i], h[j])]
        exp[group] = out
    assert max(abs(a - b) for ra, rb in zip(exp["src"], exp["dst"]) for a, b in zip(ra, rb)) > 0.05
    W, att, b = torch.eye(2, dtype=torch.float64), T(*(a_dst + a_src)).view(1, 1, 4), T(0.1, -0.2).view(-1)
    for group in ("src", "dst"):
        got = oracle.gat_conv(T(*h), P3, W, att, b, group).tolist()
        for g_row, e_row in zip(got, exp[group]):
            assert g_row == pytest.approx(e_row, abs=1e-12)


def test_scatter_mean_clamps_empty_segments():
    x = T([2.0], [4.0], [9.0])
    idx = torch.tensor([0, 0, 2])
    assert oracle.scatter_mean(x, idx, 4).view(-1).tolist() == [3.0, 0.0, 9.0, 0.0]
    assert oracle.scatter_add(x, idx, 3).view(-1).tolist() == [6.0, 0.0, 9.0]


def test_readout_multi_scale_layout():
    batch = torch.tensor([0, 0, 1])
    a1, a2 = T([1.0], [3.0], [5.0]), T([10.0, 0.0], [20.0, 2.0], [7.0, 7.0])
    out = oracle.node_aggregation([a1, a2], batch, 2, "avg_pool")
    assert out.tolist() == [[2.0, 15.0, 1.0], [5.0, 7.0, 7.0]]
    with pytest.raises(NotImplementedError):
        oracle.node_aggregation([a1], batch, 2, "max")


def test_gmn_match_by_hand():
    e = math.e
    x = T([1.0, 0.0], [1.0, 0.0], [0.0, 1.0])          # g1 = {0}, g2 = {1, 2}
    u = oracle.gmn_match(x, [(0, 1), (1, 3)])
    # node 0 vs g2: sims [1, 0] -> softmax [e, 1]/(e+1)
    assert u[0].tolist() == pytest.approx([1 - e / (e + 1), -1 / (e + 1)], abs=1e-12)
    # nodes of g2 vs g1 (single node): softmax over one element = 1 -> u = x - x0
    assert u[1].tolist() == [0.0, 0.0] and u[2].tolist() == [-1.0, 1.0]
    # scale invariance of the cosine: scaling g2's rows changes values, not attention
    x2 = x.clone(); x2[1] *= 5
    u2 = oracle.gmn_match(x2, [(0, 1), (1, 3)])
    assert u2[0].tolist() == pytest.approx([1 - 5 * e / (e + 1), -1 / (e + 1)], abs=1e-12)


def test_gmn_message_sum_restructuring_identity():
    """sum_e (W2 relu(W1 [x_i; x_j] + b1) + b2) == W2 sum_e relu(.) + deg_i b2"""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 2, generator=g, dtype=torch.float64)
    lin = [(torch.randn(4, 4, generator=g, dtype=torch.float64), torch.randn(4, generator=g, dtype=torch.float64))
           for _ in range(2)]
    ref = oracle.gmn_message_sum(x, P3, lin)
    (w1, b1), (w2, b2) = lin
    P, Q = x @ w1[:, :2].t() + b1, x @ w1[:, 2:].t()
    H = oracle.scatter_add(torch.relu(P[P3[0]] + Q[P3[1]]), P3[0], 3)
    deg = torch.tensor([1.0, 2.0, 1.0], dtype=torch.float64)
    assert torch.allclose(ref, H @ w2.t() + deg.view(-1, 1) * b2, atol=1e-12)


def test_block_diagonal_merge_and_independence():
    tri = torch.tensor([[0, 0, 1, 1, 2, 2], [1, 2, 0, 2, 0, 1]])
    xa, xb = T([1.0], [2.0], [3.0]), T([10.0], [20.0], [30.0])
    m = oracle.merge_graphs([(xa, tri), (xb, tri)])
    assert m["ind_list"] == [(0, 3), (3, 6)] and m["edge_ind_list"] == [(0, 6), (6, 12)]
    assert m["batch"].tolist() == [0, 0, 0, 1, 1, 1]
    assert m["edge_index"][:, 6:].tolist() == (tri + 3).tolist()
    w = (T([1.0]), T(0.0).view(1), T([1.0]), T(0.0).view(1))
    both = oracle.gin_conv(m["x"], m["edge_index"], *w, act="relu")
    assert both[:3].view(-1).tolist() == [6.0, 6.0, 6.0] and both[3:].view(-1).tolist() == [60.0, 60.0, 60.0]


def test_mlp_order_linear_bn_act_and_bare_last_layer():
    x = T([1.0, -1.0])
    lins = [(T([1.0, 0.0], [0.0, 1.0]), T(0.0, 0.0).view(-1)), (T([1.0, 1.0]), T(0.5).view(1))]
    bn = [(T(2.0, 2.0).view(-1), T(1.0, 1.0).view(-1), T(0.0, 0.0).view(-1), T(1.0, 1.0).view(-1) - 1e-5)]
    # hidden: relu(2*x + 1) = [3, 0]; last: sum + 0.5, NO activation
    assert oracle.mlp(x.view(1, 2), lins, "relu", bn).item() == pytest.approx(3.5, abs=1e-9)
    assert oracle.mlp(-x.view(1, 2) * 3, lins, "relu", None).item() == pytest.approx(3.5, abs=1e-12)


def test_link_pred_normalises_then_scores():
    z = T([3.0, 4.0], [0.0, 2.0], [0.0, 0.0])
    p = oracle.link_pred(z, torch.tensor([0, 2]), torch.tensor([1, 1]), type="dot_product")
    assert p.tolist() == pytest.approx([1 / (1 + math.exp(-0.8)), 0.5], abs=1e-12)   # zero row stays zero


def test_batch_norm_eval_formula():
    x = T([1.0, 2.0])
    out = oracle.batch_norm_eval(x, T(2.0, 3.0).view(-1), T(0.5, -1.0).view(-1), T(1.0, 0.0).view(-1),
                                 T(4.0, 1.0).view(-1), eps=0.0)
    assert out.view(-1).tolist() == [0.5, 5.0]


def test_spec_parser_and_unknown_names():
    name, info = oracle.parse_spec("NodeEmbedding:type=gin,input_dim=64,output_dim=64,act=identity,bn=True,normalize=False")
    assert name == "NodeEmbedding" and info["act"] == "identity" and len(info) == 6
    assert oracle.parse_spec("LoadInteractionLayer") == ("LoadInteractionLayer", {})
    with pytest.raises(ValueError):
        oracle.apply_act(T([1.0]), "gelu")
    with pytest.raises(ValueError):
        oracle.run_layers(["Nope:x=1"], {}, {"x": T([1.0])}, {})
