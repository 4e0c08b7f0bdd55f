"""GPU parity of the training step (src/train.py:177-182): gradients produced by the
libblgnn backward kernels against torch.autograd on the CPU oracle in fp64 -- op level, then the whole
lower-level GIN model (loss, every parameter gradient, running statistics, a short SGD trajectory)."""

import numpy as np
import pytest
import torch

import blgnn
import oracle
from blgnn import autograd as ag
from blgnn import ops
from helpers import TOL, max_err, build, layer_specs, state_dict_cpu, plain_batch, oracle_flags

pytestmark = pytest.mark.gpu


def _leaf(t, device=None):
    t = t.clone().to(device) if device else t.clone().double()
    return t.requires_grad_(True)


@pytest.mark.parametrize("M,K,N,act,grid", [(1000, 64, 64, "relu", True), (37001, 128, 128, "relu", True),
                                            (37001, 128, 128, "identity", False), (513, 40, 128, "identity", False),
                                            (300, 256, 1, "sigmoid", False), (64, 32, 16, "tanh", False)])
def test_linear_backward(device, M, K, N, act, grid):
    """``grid``: operands on a dyadic grid (every product and sum exact in fp32, pre-activations never 0), so the
    relu mask is the same on both sides -- with random data a handful of the M*N pre-activations land within fp32
    rounding of zero and the two masks differ there, which is the function's discontinuity, not a kernel error."""
    g = torch.Generator().manual_seed(M + N)
    if grid:
        x = torch.randint(-4, 5, (M, K), generator=g).float() / 8
        w = torch.randint(-4, 5, (N, K), generator=g).float() / 8
        b = torch.randint(-4, 5, (N,), generator=g).float() / 8 + 1.0 / 128
        dy = torch.randint(-4, 5, (M, N), generator=g).float() / 8
    else:
        x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    ref = oracle.ops.apply_act(xr @ wr.t() + br, act)
    ref.backward(dy.double())
    xd, wd, bd = _leaf(x, device), _leaf(w, device), _leaf(b, device)
    got = ag.linear(xd, wd, bd, act)
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL
    assert max_err(wd.grad, wr.grad) < TOL         # reduction over M rows: scaled by the largest entry
    assert max_err(bd.grad, br.grad) < TOL


@pytest.mark.parametrize("M,N", [(37001, 128), (5, 64), (1025, 256)])
def test_batch_norm_train_forward_backward_and_running_stats(device, M, N):
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, N, generator=g) * 2.0 + 0.5
    dy = torch.randn(M, N, generator=g)
    ref_bn = torch.nn.BatchNorm1d(N).double()
    bn = torch.nn.BatchNorm1d(N).to(device)
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(N, generator=g) + 0.5); ref_bn.bias.copy_(torch.randn(N, generator=g))
        bn.weight.copy_(ref_bn.weight.float()); bn.bias.copy_(ref_bn.bias.float())
    xr, xd = _leaf(x), _leaf(x, device)
    ref = ref_bn(xr)
    ref.backward(dy.double())
    got = ag.batch_norm_train(xd, bn)
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL
    assert max_err(bn.weight.grad, ref_bn.weight.grad) < TOL
    assert max_err(bn.bias.grad, ref_bn.bias.grad) < TOL
    assert max_err(bn.running_mean, ref_bn.running_mean) < TOL
    assert max_err(bn.running_var, ref_bn.running_var) < TOL
    assert int(bn.num_batches_tracked) == 1


def test_readout_normalise_gather_and_aggregation_backward(device):
    ds = blgnn.make_synthetic(40, 200, seed=3, device=device)
    pairs = ds.pair_gids[:32]
    bd = blgnn.BatchData(pairs, ds, is_train=False)
    m = bd.merge_data['merge']
    n, G = m.x.shape[0], len(bd.merge_data['ind_list'])
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(n, d, generator=g) for d in (64, 32)]
    batch, ei = m.batch.cpu(), m.edge_index.cpu()
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    ids1, ids2 = ids[:, 0].contiguous(), ids[:, 1].contiguous()
    w = torch.randn(ids.shape[0], 2 * 96, generator=g)

    def chain(xs_, spmm, pool, norm, gather):
        agg = [spmm(x) for x in xs_]
        z = norm(pool(agg))
        return (gather(z) * w.to(z.device).to(z.dtype)).sum()

    xr = [_leaf(x) for x in xs]
    ref = chain(xr,
                lambda x: 1.25 * x + oracle.ops.scatter_add(x[ei[1]], ei[0], n),
                lambda a: oracle.ops.node_aggregation(a, batch, G, "avg_pool"),
                oracle.ops.l2_normalize,
                lambda z: torch.cat([z[ids1], z[ids2]], dim=1))
    ref.backward()
    xd = [_leaf(x, device) for x in xs]
    seg, csr = m.segments(), m.csr()
    got = chain(xd,
                lambda x: ag.spmm_sum(x, csr, 1.25),
                lambda a: ag.segment_reduce_multi(seg, a, "mean"),
                ag.l2_normalize,
                lambda z: ag.pair_gather(z, ids1.to(device), ids2.to(device)))
    got.backward()
    assert max_err(got, ref) < TOL
    for a, b in zip(xd, xr):
        assert max_err(a.grad, b.grad) < TOL


def _llgnn(device, D=64, n_mol=60, n_pairs=400, B=48, seed=5):
    ds = blgnn.make_synthetic(n_mol, n_pairs, seed=seed, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gin", D_lower=D, device=device)
    model, data = build(flags, ds, seed=seed + 2)
    specs = layer_specs(flags)
    rng = np.random.default_rng(seed)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=B, replace=False)]
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, rng=np.random.default_rng(seed))
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    return ds, flags, model, specs, bd, ids


def _oracle_state(model):
    """fp64 copy of the state_dict; parameters (not buffers) become autograd leaves"""
    params = {k for k, _ in model.named_parameters()}
    return {k: (v.double().requires_grad_(k in params) if v.is_floating_point() else v)
            for k, v in state_dict_cpu(model).items()}


def _oracle_loss(specs, sd, plain, of, labels):
    acts = oracle.run_layers(specs, sd, plain, dict(of, train=True))
    preds = acts[-1]
    return torch.nn.functional.binary_cross_entropy(preds.view(-1), labels), acts


@pytest.mark.parametrize("shared", [False, True])
def test_prelu_forward_backward_trained_slopes(device, shared):
    """nn.PReLU(out_dim) / nn.PReLU() as an autograd node (create_act: model/layers_util.py:63-64): y, dz and the
    slope gradient against fp64 autograd; slopes of both signs and zero (the derivative tests the sign of the kept
    pre-activation, not of the output); ragged shape"""
    M, N = 2053, 70
    g = torch.Generator().manual_seed(N + shared)
    z, dy = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    a = torch.randn(1 if shared else N, generator=g) * 0.6
    if not shared:
        a[3] = 0.0
    zr, ar = _leaf(z), _leaf(a)
    ref = torch.nn.functional.prelu(zr, ar)
    ref.backward(dy.double())
    zd, ad = _leaf(z, device), _leaf(a, device)
    got = ag.prelu(zd, ad)
    got.backward(dy.to(device))
    assert torch.equal(got.detach().cpu(), torch.nn.functional.prelu(z, a))
    assert max_err(got, ref) < TOL
    assert max_err(zd.grad, zr.grad) < TOL
    assert ad.grad.shape == a.shape
    assert max_err(ad.grad, ar.grad) < TOL


@pytest.mark.parametrize("lower", ["gin", "gat", "gcn"])
def test_llgnn_prelu_training_step_gradients(device, lower):
    """A lower-level model whose NodeEmbedding layers carry act=prelu (reachable through --layer_k; model/layers.py:24-34
    builds nn.PReLU(out_dim), in a GIN layer the same module inside the MLP and behind the conv): loss, activations and
    every gradient INCLUDING the slopes' against fp64 autograd on the oracle"""
    D = 64
    ds = blgnn.make_synthetic(50, 300, seed=11, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type=lower, D_lower=D, device=device)
    for k in range(1, flags.layer_num + 1):
        spec = getattr(flags, "layer_%d" % k)
        if spec.startswith("NodeEmbedding") and "act=relu" in spec:
            setattr(flags, "layer_%d" % k, spec.replace("act=relu", "act=prelu"))
    model, data = build(flags, ds, seed=13)
    gen = torch.Generator().manual_seed(3)
    slopes = [m for m in model.modules() if isinstance(m, torch.nn.PReLU)]
    assert len(slopes) == 4 and all(m.weight.numel() == D for m in slopes)
    with torch.no_grad():
        for m in slopes:
            m.weight.copy_((torch.randn(D, generator=gen) * 0.4).to(device))          # both signs
    specs = layer_specs(flags)
    rng = np.random.default_rng(5)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=40, replace=False)]
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, rng=np.random.default_rng(5))
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    of = oracle_flags(flags, ds)
    sd = _oracle_state(model)
    labels = bd.true_labels().cpu().double()
    ref_loss, acts = _oracle_loss(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of, labels)
    ref_loss.backward()
    model.train()
    loss = model(bd)
    loss.backward()
    assert max_err(loss, ref_loss) < TOL
    for got, ref in zip(model.acts[1:-1], acts[1:]):
        assert max_err(got, ref) < TOL
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers."):
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += name.endswith("act.weight")
    assert checked == 4


@pytest.mark.parametrize("D", [64, 128])
def test_llgnn_training_step_gradients(device, D):
    """loss.backward() of the LL-GNN model (src/train.py:177-182): loss, every activation and every parameter
    gradient against fp64 autograd on the oracle; BatchNorm in train mode on both sides"""
    ds, flags, model, specs, bd, ids = _llgnn(device, D)
    of = oracle_flags(flags, ds)
    sd = _oracle_state(model)
    labels = bd.true_labels().cpu().double()
    ref_loss, acts = _oracle_loss(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of, labels)
    ref_loss.backward()

    model.train()
    loss = model(bd)
    assert loss.dim() == 0 and loss.requires_grad
    loss.backward()
    assert max_err(loss, ref_loss) < TOL
    for got, ref in zip(model.acts[1:-1], acts[1:]):
        assert max_err(got, ref) < TOL
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers."):
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 5 * 6 + 4          # 5 GIN layers x (2 Linear x 2 + BN x 2) + the scorer's Linears
    # a forward in eval mode still takes the fused path and uses the running statistics just updated
    model.eval()
    with torch.no_grad():
        eval_loss = model(bd)
    assert not eval_loss.requires_grad and torch.isfinite(eval_loss)


def test_llgnn_sgd_trajectory(device):
    """ten optimizer steps on one batch: the loss follows the oracle's fp64 trajectory and goes down.  (Momentum
    SGD, whose step is linear in the gradient; Adam turns gradient entries that are zero up to rounding into
    full-size steps of either sign, so its fp32 and fp64 trajectories part after a few steps on ANY backend.)"""
    ds, flags, model, specs, bd, ids = _llgnn(device, 64, seed=9)
    of = oracle_flags(flags, ds)
    sd = _oracle_state(model)
    train_keys = [k for k, _ in model.named_parameters() if k.startswith("layers.")]
    opt_ref = torch.optim.SGD([sd[k] for k in train_keys], lr=0.02, momentum=0.9)
    opt = torch.optim.SGD([p for k, p in model.named_parameters() if k.startswith("layers.")], lr=0.02, momentum=0.9)
    labels = bd.true_labels().cpu().double()
    plain = plain_batch(bd, ds, torch.float64, with_pairs=ids)
    model.train()
    got, ref = [], []
    for _ in range(10):
        opt_ref.zero_grad()
        rl, _a = _oracle_loss(specs, sd, plain, of, labels)
        rl.backward()
        opt_ref.step()
        ref.append(float(rl.detach()))
        opt.zero_grad()
        loss = model(bd)
        loss.backward()
        opt.step()
        got.append(float(loss.detach()))
    assert got[-1] < got[0]
    assert max(abs(a - b) for a, b in zip(got, ref)) < 1e-4, (got, ref)


@pytest.mark.parametrize("group", ["src", "dst"])
@pytest.mark.parametrize("n,deg,d,act", [(300, 6, 64, "relu"), (90, 40, 128, "identity"), (7, 2, 32, "tanh")])
def test_gat_conv_backward(device, group, n, deg, d, act):
    """GATConv backward (both softmax groupings; existing self loops, isolated nodes, rows of very different
    length) against fp64 autograd on the oracle: dx, dW, datt, dbias"""
    from test_gpu_ops import rand_graph
    ei = rand_graph(n, deg, seed=n + deg, self_loops=True, isolated=2)
    g = torch.Generator().manual_seed(d)
    K = 48
    x, w = torch.randn(n, K, generator=g), torch.randn(K, d, generator=g) / K ** 0.5
    att, b = torch.randn(1, 1, 2 * d, generator=g) * 0.3, torch.randn(d, generator=g) * 0.1
    if act == "relu":                      # keep pre-activations away from the kink (see test_linear_backward)
        b = b + 3.0
    dy = torch.randn(n, d, generator=g)
    xr, wr, ar, br = _leaf(x), _leaf(w), _leaf(att), _leaf(b)
    ref = oracle.ops.apply_act(oracle.ops.gat_conv(xr, ei, wr, ar, br, group), act)
    ref.backward(dy.double())
    xd, wd, ad, bd = _leaf(x, device), _leaf(w, device), _leaf(att, device), _leaf(b, device)
    csr = ops.csr_from_edge_index(ei.to(device), n)
    got = ag.gat_conv(xd, wd, ad, bd, csr, group=group, act=act)
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL
    assert max_err(wd.grad, wr.grad) < TOL
    assert max_err(ad.grad, ar.grad) < TOL
    assert max_err(bd.grad, br.grad) < TOL


def test_gcn_conv_backward(device):
    from test_gpu_ops import rand_graph
    n, K, d = 400, 40, 64
    ei = rand_graph(n, 5, seed=3, self_loops=True, isolated=3)
    g = torch.Generator().manual_seed(2)
    x, w, b = torch.randn(n, K, generator=g), torch.randn(K, d, generator=g) / K ** 0.5, torch.randn(d, generator=g)
    dy = torch.randn(n, d, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    ref = torch.tanh(oracle.ops.gcn_conv(xr, ei, wr, br))
    ref.backward(dy.double())
    xd, wd, bd = _leaf(x, device), _leaf(w, device), _leaf(b, device)
    got = ag.gcn_conv(xd, wd, bd, ops.csr_from_edge_index(ei.to(device), n), act="tanh")
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    for a, r in ((xd, xr), (wd, wr), (bd, br)):
        assert max_err(a.grad, r.grad) < TOL


@pytest.mark.parametrize("lower", ["gin", "gat", "gcn"])
def test_bignn_training_step_gradients(device, lower):
    """The bi-level model trained end to end as src/train.py:75-108,177-182 does it: the molecular sweep runs WITH
    gradients, its pooled rows become the interaction graph's init_x, the interaction-level GAT layers and the pair
    scorer produce the loss, and backward reaches every parameter of both levels.  Loss + all parameter gradients
    against fp64 autograd on the oracle (train-mode BatchNorm on both sides)."""
    from blgnn.forward import get_initial_embd, all_molecule_pairs
    D = 64
    ds = blgnn.make_synthetic(70, 500, seed=12, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type=lower, higher_level_gnn_type="gat",
                             D_lower=D, D_higher=D, device=device)
    model, data = build(flags, ds, seed=14)
    specs = layer_specs(flags)
    rng = np.random.default_rng(2)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=40, replace=False)]
    # ---- ours
    model.train()
    get_initial_embd(data, model)
    ds.init_interaction_graph_embds(device)
    assert ds.interaction_combo_nxgraph.init_x.requires_grad
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, rng=np.random.default_rng(3))
    model.use_layers = 'higher_layers'
    loss = model(bd)
    loss.backward()
    # ---- oracle: one layer list, the sweep's batch + the pair rows
    lower_b = blgnn.BatchData(all_molecule_pairs(ds.gs_map.keys()), ds, is_train=False, ignore_pairs=True)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    sd = _oracle_state(model)
    of = dict(oracle_flags(flags, ds), train=True)
    acts = oracle.run_layers(specs, sd, plain_batch(lower_b, ds, torch.float64, with_pairs=ids), of)
    ref_loss = torch.nn.functional.binary_cross_entropy(acts[-1].view(-1), bd.true_labels().cpu().double())
    ref_loss.backward()
    assert max_err(loss, ref_loss) < TOL
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers."):
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 5 * {"gin": 6, "gat": 5, "gcn": 4}[lower] + 3 * 5 + 4


def test_bignn_training_two_chunk_sweep_gradients(device):
    """src/train.py:48-72 sweeps the molecules in chunks of FLAGS.batch_size pairs and writes every chunk's pooled
    rows into init_x in place, so ALL chunks stay in the autograd graph (and train-mode BatchNorm takes per-chunk
    statistics).  A two-chunk sweep here: loss and every parameter gradient against fp64 autograd on the oracle
    run the same way -- a chunk whose rows were detached would show up as wrong lower-level gradients."""
    from blgnn.forward import get_initial_embd, all_molecule_pairs
    D = 64
    ds = blgnn.make_synthetic(70, 500, seed=21, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=D, D_higher=D, device=device)
    model, data = build(flags, ds, seed=22)
    specs = layer_specs(flags)
    rng = np.random.default_rng(4)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=40, replace=False)]
    allp = all_molecule_pairs(ds.gs_map.keys())
    step = (len(allp) + 1) // 2
    assert step < len(allp)
    # ---- ours: two chunks, gradients through both
    model.train()
    get_initial_embd(data, model, chunk_pairs=step)
    ds.init_interaction_graph_embds(device)
    assert ds.interaction_combo_nxgraph.init_x.requires_grad
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, rng=np.random.default_rng(3))
    model.use_layers = 'higher_layers'
    loss = model(bd)
    loss.backward()
    # ---- oracle: the lower slice once per chunk (in-place init_x rows, as the reference), then the upper slice
    n_lower = 1 + max(k for k, sp in enumerate(specs) if sp.startswith("NodeAggregation"))
    sd = _oracle_state(model)
    of = dict(oracle_flags(flags, ds), train=True)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    init_x, plain = None, None
    for i in range(0, len(allp), step):
        chunk = blgnn.BatchData(allp[i:i + step], ds, is_train=False, ignore_pairs=True)
        plain = plain_batch(chunk, ds, torch.float64, with_pairs=ids)
        plain["init_x"] = init_x
        oracle.run_layers(specs[:n_lower], sd, plain, of)
        init_x = plain["init_x"]
    acts = oracle.run_layers(specs[n_lower:], sd, plain, of, first=n_lower, x0=init_x)
    ref_loss = torch.nn.functional.binary_cross_entropy(acts[-1].view(-1), bd.true_labels().cpu().double())
    ref_loss.backward()
    assert max_err(loss, ref_loss) < TOL
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers."):
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 5 * 6 + 3 * 5 + 4
    # a second sweep starts from values: it must not chain onto the first sweep's (already freed) graph
    get_initial_embd(data, model, chunk_pairs=step)
    ds.init_interaction_graph_embds(device)
    model.use_layers = 'higher_layers'
    model(bd).backward()


def test_drugcombo_bignn_training_gradients(device):
    """DrugCombo variant of the bi-level model: the interaction level sums one GATConv per edge type
    (NodeModelAggrByEdge('gat'), model/layers_meta.py:70,74-79) and the scorer emits one logit per label; gradients
    of a fixed linear functional of the logits w.r.t. every parameter against the oracle"""
    from blgnn.forward import get_initial_embd, all_molecule_pairs
    ds = blgnn.make_synthetic(70, 500, seed=8, device=device, num_edge_types=2, name="drugcombo_syn")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset="drugcombo", lower_level_gnn_type="gin",
                             device=device)
    model, data = build(flags, ds, seed=11, num_hyper_edge_feat=3)
    specs = layer_specs(flags)
    rng = np.random.default_rng(5)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=32, replace=False)]
    model.train()
    get_initial_embd(data, model)
    ds.init_interaction_graph_embds(device)
    bd = blgnn.BatchData(pairs, ds, is_train=False)
    model.use_layers = 'higher_layers'
    model.acts = [bd.merge_data['merge'].x]
    for layer in model._selected()[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    logits = model.acts[-1]
    assert logits.shape == (32, 3) and logits.requires_grad
    wgt = torch.randn(32, 3, generator=torch.Generator().manual_seed(0))
    (logits * wgt.to(device)).sum().backward()

    lower_b = blgnn.BatchData(all_molecule_pairs(ds.gs_map.keys()), ds, is_train=False, ignore_pairs=True)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    sd = _oracle_state(model)
    acts = oracle.run_layers(specs, sd, plain_batch(lower_b, ds, torch.float64, with_pairs=ids),
                             dict(oracle_flags(flags, ds), train=True))
    assert max_err(logits, acts[-1]) < TOL
    (acts[-1] * wgt.double()).sum().backward()
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers.") or sd[name].grad is None:
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 30 + 3 * 3 * 2


def test_train_mode_batch_norm_synchronised_over_two_shards(device):
    """data-parallel train-mode BatchNorm (autograd.set_batch_norm_sync): each of two shards of one batch is processed
    with a reduce hook that adds what the OTHER shard would have contributed to the all-reduce (computed with the same
    kernels on the other shard's rows); every shard must normalise with the whole batch's statistics and produce the
    whole batch's input gradient"""
    g = torch.Generator().manual_seed(4)
    M, N = 3001, 128
    x_all, dy_all = torch.randn(M, N, generator=g) * 1.5 + 0.3, torch.randn(M, N, generator=g)
    ref_bn = torch.nn.BatchNorm1d(N).double()
    with torch.no_grad():
        ref_bn.weight.copy_(torch.rand(N, generator=g) + 0.5); ref_bn.bias.copy_(torch.randn(N, generator=g))
    xr = _leaf(x_all)
    ref = ref_bn(xr)
    ref.backward(dy_all.double())
    bounds = [(0, 1800), (1800, M)]

    def other_rank(x_o, dy_o):
        st = {"k": 0}

        def reduce(t):
            k = st["k"]; st["k"] += 1
            if k == 0:                                     # [sum_x | rows]
                t[:N] += ops.col_reduce(x_o, "sum"); t[N] += float(x_o.shape[0])
                st["count"] = float(t[N]); st["mean"] = (t[:N] / t[N]).contiguous()
            elif k == 1:                                   # centred squares
                t += ops.col_reduce(x_o, "sqdev", mean=st["mean"])
                st["rstd"] = torch.rsqrt(t / st["count"] + ref_bn.eps)
            else:                                          # [dbeta ; dgamma]
                t[0] += ops.col_reduce(dy_o, "sum")
                t[1] += ops.col_reduce(dy_o, "dot_norm", b=x_o, mean=st["mean"], rstd=st["rstd"])
            return t
        return reduce

    grads = []
    for r, (lo, hi) in enumerate(bounds):
        olo, ohi = bounds[1 - r]
        bn = torch.nn.BatchNorm1d(N).to(device)
        with torch.no_grad():
            bn.weight.copy_(ref_bn.weight.float()); bn.bias.copy_(ref_bn.bias.float())
        ag.set_batch_norm_sync(other_rank(x_all[olo:ohi].to(device), dy_all[olo:ohi].to(device)))
        try:
            x = _leaf(x_all[lo:hi], device)
            y = ag.batch_norm_train(x, bn)
            y.backward(dy_all[lo:hi].to(device))
        finally:
            ag.set_batch_norm_sync(None)
        assert max_err(y, ref[lo:hi]) < TOL
        assert max_err(x.grad, xr.grad[lo:hi]) < TOL
        assert max_err(bn.running_var, ref_bn.running_var) < TOL
        assert max_err(bn.running_mean, ref_bn.running_mean) < TOL
        grads.append((bn.weight.grad, bn.bias.grad))
    assert max_err(grads[0][0] + grads[1][0], ref_bn.weight.grad) < TOL      # local sums: the wrapper adds them up
    assert max_err(grads[0][1] + grads[1][1], ref_bn.bias.grad) < TOL


def test_edge_relu_sum_and_gmn_match_backward(device):
    """the two GMN-specific operators: gradients of sum_e relu(P_i + Q_j) and of the cross-graph attention (pairs with
    small, tile-edge, > 64- and > 128-node graphs, an empty graph) against fp64 autograd on the oracle"""
    from test_gpu_ops import molecule_graph
    ei, _, n = molecule_graph(30, 5)
    g = torch.Generator().manual_seed(8)
    d = 48
    p, q = torch.randn(n, d, generator=g) + 0.2, torch.randn(n, d, generator=g)
    ds_ = torch.randn(n, d, generator=g)
    pr, qr = _leaf(p), _leaf(q)
    ref = oracle.ops.scatter_add(torch.relu(pr[ei[0]] + qr[ei[1]]), ei[0], n)
    ref.backward(ds_.double())
    pd, qd = _leaf(p, device), _leaf(q, device)
    got = ag.edge_relu_sum(pd, qd, ops.csr_from_edge_index(ei.to(device), n))
    got.backward(ds_.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(pd.grad, pr.grad) < TOL and max_err(qd.grad, qr.grad) < TOL

    sizes = [5, 7, 33, 31, 64, 65, 130, 20, 0, 9, 1, 1]
    off = np.concatenate([[0], np.cumsum(sizes)])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    D = 64
    x, du = torch.randn(int(off[-1]), D, generator=g), torch.randn(int(off[-1]), D, generator=g)
    xr = _leaf(x)
    ref = oracle.ops.gmn_match(xr, ind)
    ref.backward(du.double())
    xd = _leaf(x, device)
    got = ag.gmn_match(xd, torch.as_tensor(off, dtype=torch.int32, device=device))
    got.backward(du.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL


@pytest.mark.parametrize("D", [256, 512, 72])
def test_gmn_match_backward_staged_and_global_paths(device, D):
    """the attention backward with one graph of the pair staged in LDS (graphs up to 64 rows at d = 256, 37 at d = 512)
    next to pairs that take the global-memory path in the same launch (a larger graph), an empty graph, one-node graphs,
    a width that is not a multiple of 64 -- dx against fp64 autograd on the oracle"""
    sizes = [5, 7, 33, 31, 64, 64, 65, 12, 37, 38, 130, 20, 0, 9, 1, 1, 29, 30]
    off = np.concatenate([[0], np.cumsum(sizes)])
    ind = [(int(off[i]), int(off[i + 1])) for i in range(len(sizes))]
    g = torch.Generator().manual_seed(D)
    x, du = torch.randn(int(off[-1]), D, generator=g), torch.randn(int(off[-1]), D, generator=g)
    xr = _leaf(x)
    ref = oracle.ops.gmn_match(xr, ind)
    ref.backward(du.double())
    xd = _leaf(x, device)
    got = ag.gmn_match(xd, torch.as_tensor(off, dtype=torch.int32, device=device))
    got.backward(du.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL


def test_mhcaddi_gmn_training_step_gradients(device):
    """the GMN model (lower_level_gmn: GIN + GMNPropagator layers, pair mode) under loss.backward(): loss and every
    parameter gradient against fp64 autograd on the oracle, train-mode BatchNorm in the layers and inside f_node"""
    ds = blgnn.make_synthetic(40, 200, seed=6, device=device)
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=64, device=device)
    model, data = build(flags, ds, seed=9)
    specs = layer_specs(flags)
    rng = np.random.default_rng(1)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=16, replace=False)]
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, unique_graphs=False,
                         rng=np.random.default_rng(2))
    B = len(bd.batch_gids)
    ids = torch.arange(2 * B).reshape(-1, 2)
    sd = _oracle_state(model)
    of = dict(oracle_flags(flags, ds), train=True)
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    ref_loss = torch.nn.functional.binary_cross_entropy(acts[-1].view(-1), bd.true_labels().cpu().double())
    ref_loss.backward()
    model.train()
    loss = model(bd)
    loss.backward()
    assert max_err(loss, ref_loss) < TOL
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers.") or sd[name].grad is None:
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 3 * 6 + 3 * 10


def test_gru_gates_backward(device):
    """the GRU gate arithmetic (torch.nn.GRUCell) and its backward kernel against fp64 autograd: dgi, dgh, dh"""
    n, D = 777, 48
    g = torch.Generator().manual_seed(4)
    gi, gh, h, dy = (torch.randn(n, 3 * D, generator=g), torch.randn(n, 3 * D, generator=g), torch.randn(n, D, generator=g),
                     torch.randn(n, D, generator=g))
    gir, ghr, hr = _leaf(gi), _leaf(gh), _leaf(h)
    r = torch.sigmoid(gir[:, :D] + ghr[:, :D])
    z = torch.sigmoid(gir[:, D:2 * D] + ghr[:, D:2 * D])
    c = torch.tanh(gir[:, 2 * D:] + r * ghr[:, 2 * D:])
    ref = (1 - z) * c + z * hr
    ref.backward(dy.double())
    gid, ghd, hd = _leaf(gi, device), _leaf(gh, device), _leaf(h, device)
    got = ag.gru_gates(gid, ghd, hd)
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    for a, b in ((gid, gir), (ghd, ghr), (hd, hr)):
        assert max_err(a.grad, b.grad) < TOL


def test_mhcaddi_gmn_gru_training_step_gradients(device):
    """the GMN model with f_node='GRU' (model/layers_gmn.py:27-30,48-50) under loss.backward(): loss and every
    parameter gradient (GRUCell weight_ih / weight_hh / bias_ih / bias_hh included) against fp64 autograd on the oracle"""
    ds = blgnn.make_synthetic(40, 200, seed=6, device=device)
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=64, device=device, gmn_f_node="GRU")
    model, data = build(flags, ds, seed=9)
    assert isinstance(model.layers[1].f_node, torch.nn.GRUCell)
    specs = layer_specs(flags)
    rng = np.random.default_rng(1)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=16, replace=False)]
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, unique_graphs=False,
                         rng=np.random.default_rng(2))
    B = len(bd.batch_gids)
    ids = torch.arange(2 * B).reshape(-1, 2)
    sd = _oracle_state(model)
    of = dict(oracle_flags(flags, ds), train=True)
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    ref_loss = torch.nn.functional.binary_cross_entropy(acts[-1].view(-1), bd.true_labels().cpu().double())
    ref_loss.backward()
    model.train()
    loss = model(bd)
    loss.backward()
    assert max_err(loss, ref_loss) < TOL
    checked = gru = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers.") or sd[name].grad is None:
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
        gru += ".f_node.weight_" in name or ".f_node.bias_" in name
    assert gru == 3 * 4 and checked >= 3 * 6 + 3 * 4


@pytest.mark.parametrize("M,O,I", [(37001, 128, 128), (1, 5, 7), (1025, 130, 40), (2049, 64, 300), (16, 1, 1), (0, 8, 8)])
def test_grad_weight_and_column_reductions_ragged_shapes(device, M, O, I):
    """the row-chunked weight-gradient reduction and the deterministic column reductions: chunk edges (1 024 / 512
    rows), output blocks that are not multiples of 32 / 128, a single row, no rows; strided operands"""
    g = torch.Generator().manual_seed(M + O + I)
    gw, xw = torch.randn(M, O + 3, generator=g), torch.randn(M, I + 5, generator=g)
    gd, xd = gw.to(device)[:, 1:1 + O], xw.to(device)[:, 2:2 + I]           # column slices of wider buffers
    ref = gw[:, 1:1 + O].double().t() @ xw[:, 2:2 + I].double()
    got = ops.grad_weight(gd, xd)
    assert got.shape == (O, I)
    assert max_err(got, ref) < TOL
    if M > 0:
        a, b = gw[:, 1:1 + O].double(), torch.randn(M, O, generator=g).double()
        mean = a.mean(0)
        rstd = 1.0 / torch.sqrt(((a - mean) ** 2).mean(0) + 1e-5)
        bd = b.float().to(device)
        dv = lambda t: t.float().to(device).contiguous()
        assert max_err(ops.col_reduce(gd, "sum"), a.sum(0)) < TOL
        assert max_err(ops.col_reduce(gd, "dot", b=bd), (a * b).sum(0)) < TOL
        assert max_err(ops.col_reduce(gd, "sqdev", mean=dv(mean)), ((a - mean) ** 2).sum(0)) < TOL
        assert max_err(ops.col_reduce(bd, "dot_norm", b=gd, mean=dv(mean), rstd=dv(rstd)),
                       (b * (a - mean) * rstd).sum(0)) < TOL
        # bit-for-bit reproducible
        assert torch.equal(ops.grad_weight(gd, xd), got)
        assert torch.equal(ops.col_reduce(gd, "sum"), ops.col_reduce(gd, "sum"))


@pytest.mark.parametrize("M,O,I", [(37001, 128, 128), (1025, 132, 40), (2049, 64, 300), (33, 4, 8), (5000, 256, 128)])
def test_grad_weight_aligned_operands_take_the_lds_staged_kernel(device, M, O, I):
    """16-byte aligned rows with O, I multiples of 4 (the layers' own activations): the LDS-staged weight-gradient kernel
    -- row blocks of 32 with a ragged last block, output blocks cut by O / I, several 128 x 128 output blocks; against
    fp64 and bit for bit reproducible"""
    g = torch.Generator().manual_seed(M + O + I + 1)
    gw, xw = torch.randn(M, O, generator=g), torch.randn(M, I, generator=g)
    ref = gw.double().t() @ xw.double()
    gd, xd = gw.to(device), xw.to(device)
    got = ops.grad_weight(gd, xd)
    assert max_err(got, ref) < TOL
    assert torch.equal(ops.grad_weight(gd, xd), got)
    wide = torch.randn(M, I + 8, generator=g).to(device)                       # aligned column slice of a wider buffer
    assert max_err(ops.grad_weight(gd, wide[:, 4:4 + I]), gw.double().t() @ wide[:, 4:4 + I].double().cpu()) < TOL


def test_train_step_captured_equals_eager(device):
    """blgnn.TrainStep: the training iteration (molecular sweep with gradients -> init_x -> pair batch -> BCE -> backward
    -> SGD step) captured in a HIP graph and replayed on fresh batches follows the same trajectory as the identical
    iteration run eagerly: losses and every parameter / BatchNorm statistic after six batches.  The pair scatter and the GAT
    score gradients use float atomics, so two EAGER runs already differ in the last bits and training amplifies that
    (measured over several runs: 4e-7..2e-5 on the losses, 2e-6..2e-5 on the parameters at this learning rate; at lr = 0.05 two eager runs are
    1e-2 apart after six steps): the bars below are those of run-to-run variation, not bit equality"""
    import copy
    from blgnn.batch import BatchData
    ds = blgnn.make_synthetic(60, 400, seed=21, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=64, D_higher=64, device=device, batch_size=32)
    blgnn.set_flags(flags)
    torch.manual_seed(3)
    data = blgnn.TrainData(ds, device=device)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    rng = np.random.default_rng(5)
    batches = []
    while len(batches) < 7:
        pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=32, replace=False)]
        b = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=rng)
        if len(b.batch_gids) == 64:
            batches.append(b)
    results = []
    for capture in (False, True):
        torch.manual_seed(3)
        model = blgnn.Model(data).to(device)                   # same initial parameters for both runs
        model.train()
        ds.init = False
        ds.interaction_combo_nxgraph.init_x = None
        ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
        opt = torch.optim.SGD([p for k, p in model.named_parameters() if k.startswith("layers.")], lr=0.005, momentum=0.9)
        first = copy.copy(batches[0])
        ts = blgnn.TrainStep(model, data, opt, first, capture=capture, warmup=2)
        assert (ts.graph is not None) == capture
        if not capture:                                         # the eager twin trains on the first batch as often as the
            for _ in range(2):                                  # captured one's warm-up did (capture itself records, it
                ts.step(batches[0])                             # does not execute)
        losses = [float(ts.step(b)) for b in batches[1:]]
        torch.cuda.synchronize()
        results.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert abs(l0[0] - l1[0]) < 2e-6, (l0, l1)                     # same state, same batch: rounding only
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-4, (l0, l1)   # (a stale id / label buffer shows up as 1e-2)
    assert l0[-1] != l0[0]
    for k in s0:
        if s0[k].is_floating_point():
            # (not an oracle comparison: kept out of the parity log that helpers.max_err writes)
            a, b = s0[k].double().cpu(), s1[k].double().cpu()
            assert float((a - b).abs().max()) / max(1.0, float(a.abs().max())) < 1e-3, k
        else:
            assert torch.equal(s0[k], s1[k]), k


def test_train_step_replays_invalidate_derived_weight_caches(device):
    """A captured TrainStep updates parameters and BatchNorm statistics inside graph replays, which never move
    ``Tensor._version`` -- the key of every derived-tensor cache (transposed / packed weights, the eval-BN fold).  The
    reference validates between training iterations (src/train.py:152): train(replay) -> eval -> train(replay) -> eval must
    score with the CURRENT weights both times.  Checked against a fresh module (no caches) loaded with the trained
    state_dict."""
    import copy
    from blgnn.batch import BatchData
    from blgnn.forward import bilevel_forward
    ds = blgnn.make_synthetic(60, 400, seed=22, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=64, D_higher=64, device=device, batch_size=32)
    blgnn.set_flags(flags)
    torch.manual_seed(4)
    data = blgnn.TrainData(ds, device=device)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    rng = np.random.default_rng(6)
    batches = []
    while len(batches) < 5:
        pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=32, replace=False)]
        b = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=rng)
        if len(b.batch_gids) == 64:
            batches.append(b)
    model = blgnn.Model(data).to(device)
    model.train()
    opt = torch.optim.SGD([p for k, p in model.named_parameters() if k.startswith("layers.")], lr=0.05, momentum=0.9)
    ts = blgnn.TrainStep(model, data, opt, copy.copy(batches[0]), capture=True, warmup=2)
    assert ts.graph is not None
    eval_pairs = ds.pair_gids[:48]

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            preds, _ = bilevel_forward(m, data, eval_pairs)
        torch.cuda.synchronize()
        m.train()
        return preds.detach().clone()

    def fresh_twin():
        twin = blgnn.Model(data).to(device)
        twin.load_state_dict(copy.deepcopy(model.state_dict()))
        return twin

    seen = []
    for rounds in range(2):                          # train(replay) -> eval, twice
        for b in batches[1 + 2 * rounds:3 + 2 * rounds]:
            ts.step(b)
        torch.cuda.synchronize()
        got = evaluate(model)                        # the trained module: its caches saw the weights of the last eval
        want = evaluate(fresh_twin())                # no caches
        # (not an oracle comparison: kept out of the parity log)
        assert float((got - want).abs().max()) < 1e-6, rounds
        seen.append(got)
    assert float((seen[0] - seen[1]).abs().max()) > 1e-4         # the second round of replays did change the scores


def test_train_step_drugcombo_bilevel_captured_equals_eager(device):
    """blgnn.TrainStep on the DrugCombo bi-level model (src/config.py:120-123,383-410: MetaLayer interaction level = one
    GATConv per edge type summed, multi-label scorer, cross-entropy): the captured iteration follows the eager one over
    four batches (bars: run-to-run variation of the float atomics in the backward kernels, as in the DrugBank test)"""
    import copy
    from blgnn.batch import BatchData
    ds = blgnn.make_synthetic(60, 400, seed=23, device=device, num_edge_types=2, name="drugcombo_syn")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset="drugcombo", lower_level_gnn_type="gin",
                             D_lower=64, D_higher=64, device=device, batch_size=32)
    blgnn.set_flags(flags)
    torch.manual_seed(3)
    data = blgnn.TrainData(ds, device=device, num_hyper_edge_feat=3)
    ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
    rng = np.random.default_rng(5)
    batches = []
    while len(batches) < 5:
        pos = ds.pair_gids[rng.choice(len(ds.pair_gids), size=32, replace=False)]
        b = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=rng)
        if not batches or len(b.batch_gids) == len(batches[0].batch_gids):
            batches.append(b)
    results = []
    for capture in (False, True):
        torch.manual_seed(3)
        model = blgnn.Model(data).to(device)
        model.train()
        ds.init = False
        ds.interaction_combo_nxgraph.init_x = None
        ds.init_interaction_graph_feats(flags.init_embds, device, flags.d_init)
        opt = torch.optim.SGD([p for k, p in model.named_parameters() if k.startswith("layers.")], lr=0.005, momentum=0.9)
        ts = blgnn.TrainStep(model, data, opt, copy.copy(batches[0]), capture=capture, warmup=2)
        assert (ts.graph is not None) == capture
        if not capture:
            for _ in range(2):
                ts.step(batches[0])
        losses = [float(ts.step(b)) for b in batches[1:]]
        torch.cuda.synchronize()
        results.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    assert abs(l0[0] - l1[0]) < 2e-6, (l0, l1)
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-4, (l0, l1)
    assert l0[-1] != l0[0]
    for k in s0:
        if s0[k].is_floating_point():
            a, b = s0[k].double().cpu(), s1[k].double().cpu()
            assert float((a - b).abs().max()) / max(1.0, float(a.abs().max())) < 1e-3, k


@pytest.mark.parametrize("node_aggr", ["deepsets", "gmn_aggr"])
def test_llgnn_training_step_gradients_learned_readouts(device, node_aggr):
    """the DeepSets and gated GMN readouts train (model/layers_aggregation.py:45-56,78-95): loss, every activation and
    every parameter gradient (the readout's MLPs and, for gmn_aggr, their train-mode BatchNorms included) against fp64
    autograd on the oracle"""
    ds = blgnn.make_synthetic(60, 400, seed=5, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gin", D_lower=64, device=device,
                             node_aggr=node_aggr)
    model, data = build(flags, ds, seed=7)
    specs = layer_specs(flags)
    rng = np.random.default_rng(5)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=48, replace=False)]
    bd = blgnn.BatchData(pairs, ds, sampled_gids=np.unique(pairs), is_train=True, rng=np.random.default_rng(5))
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    of = oracle_flags(flags, ds)
    sd = _oracle_state(model)
    labels = bd.true_labels().cpu().double()
    ref_loss, acts = _oracle_loss(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of, labels)
    ref_loss.backward()

    model.train()
    loss = model(bd)
    assert loss.requires_grad
    loss.backward()
    assert max_err(loss, ref_loss) < TOL
    for got, ref in zip(model.acts[1:-1], acts[1:]):
        assert max_err(got, ref) < TOL
    readout = [n for n, _ in model.named_parameters() if ".agg_func." in n]
    assert len(readout) >= (8 if node_aggr == "deepsets" else 18)
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers."):
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 5 * 6 + 4 + len(readout)


def test_multi_edge_gcn_backward(device):
    """NodeModelAggrByEdge('gcn') (model/layers_meta.py:68,74-79): act(sum_k GCNConv_k(x, edges_k)), forward and every
    gradient against fp64 autograd on the oracle; and through MetaLayerWrapper in train mode"""
    import types
    from test_gpu_ops import rand_graph
    from blgnn.layers import GCNConv
    n, K, d = 350, 48, 64
    eis = [rand_graph(n, deg, seed=s, self_loops=False, isolated=2) for deg, s in ((5, 3), (9, 4), (2, 5))]
    g = torch.Generator().manual_seed(6)
    x, dy = torch.randn(n, K, generator=g), torch.randn(n, d, generator=g)
    convs = [GCNConv(K, d).to(device) for _ in eis]
    with torch.no_grad():
        for c in convs:
            c.bias.copy_(torch.randn(d, generator=g))
    xr = _leaf(x)
    pr = [(_leaf(c.weight.detach().cpu()), _leaf(c.bias.detach().cpu())) for c in convs]
    ref = torch.tanh(sum(oracle.ops.gcn_conv(xr, ei, w, b) for ei, (w, b) in zip(eis, pr)))
    ref.backward(dy.double())
    xd = _leaf(x, device)
    csrs = [ops.csr_from_edge_index(ei.to(device), n) for ei in eis]
    got = ag.multi_edge_gcn(xd, convs, csrs, act="tanh")
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    assert max_err(xd.grad, xr.grad) < TOL
    for c, (w, b) in zip(convs, pr):
        assert max_err(c.weight.grad, w.grad) < TOL
        assert max_err(c.bias.grad, b.grad) < TOL


def test_drugcombo_gcn_multi_edge_training_gradients(device):
    """the DrugCombo bi-level model with node_model='gcn_multi_edge_aggr' trains: gradients of a linear functional of
    the logits w.r.t. every parameter against the oracle (the 'gat' twin: test_drugcombo_bignn_training_gradients)"""
    from blgnn.forward import get_initial_embd, all_molecule_pairs
    ds = blgnn.make_synthetic(70, 500, seed=8, device=device, num_edge_types=2, name="drugcombo_syn")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset="drugcombo", lower_level_gnn_type="gin",
                             device=device, multi_edge_node_model="gcn")
    model, data = build(flags, ds, seed=11, num_hyper_edge_feat=3)
    specs = layer_specs(flags)
    rng = np.random.default_rng(5)
    pairs = ds.pair_gids[rng.choice(len(ds.pair_gids), size=32, replace=False)]
    model.train()
    get_initial_embd(data, model)
    ds.init_interaction_graph_embds(device)
    bd = blgnn.BatchData(pairs, ds, is_train=False)
    model.use_layers = 'higher_layers'
    model.acts = [bd.merge_data['merge'].x]
    for layer in model._selected()[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    logits = model.acts[-1]
    assert logits.shape == (32, 3) and logits.requires_grad
    wgt = torch.randn(32, 3, generator=torch.Generator().manual_seed(0))
    (logits * wgt.to(device)).sum().backward()

    lower_b = blgnn.BatchData(all_molecule_pairs(ds.gs_map.keys()), ds, is_train=False, ignore_pairs=True)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    sd = _oracle_state(model)
    acts = oracle.run_layers(specs, sd, plain_batch(lower_b, ds, torch.float64, with_pairs=ids),
                             dict(oracle_flags(flags, ds), train=True))
    assert max_err(logits, acts[-1]) < TOL
    (acts[-1] * wgt.double()).sum().backward()
    checked = 0
    for name, p in model.named_parameters():
        if name not in sd or not name.startswith("layers.") or sd[name].grad is None:
            continue
        assert p.grad is not None, name
        assert max_err(p.grad, sd[name].grad) < TOL, name
        checked += 1
    assert checked >= 30 + 3 * 2 * 2


def test_link_pred_dot_product_trains(device):
    """LinkPred(type='dot_product') (model/layers_link_pred.py:65-66) in train mode: scores and the gradient w.r.t. the
    graph embeddings against fp64 autograd on the oracle"""
    import types
    from blgnn.layers_link_pred import LinkPred
    g = torch.Generator().manual_seed(4)
    z = torch.randn(90, 96, generator=g)
    ids = torch.randint(0, 90, (300, 2), generator=g)
    wgt = torch.randn(300, generator=g)
    zr = _leaf(z)
    ref = oracle.link_pred(zr, ids[:, 0], ids[:, 1], None, "dot_product")
    (ref.view(-1) * wgt.double()).sum().backward()
    lp = LinkPred("dot_product", 96, 1).train()
    bd = types.SimpleNamespace(_pair_row_ids=(ids[:, 0].contiguous().to(device), ids[:, 1].contiguous().to(device)))
    zd = _leaf(z, device)
    got = lp(zd, bd, None)
    assert got.requires_grad
    (got.view(-1) * wgt.to(device)).sum().backward()
    assert max_err(got.view(-1), ref.view(-1)) < TOL
    assert max_err(zd.grad, zr.grad) < TOL


def test_gcn_conv_edge_weights_forward_backward(device):
    """GCNConv with per-edge weights (model/layers.py:48-52 hands the merge's edge_attr to the conv): forward and every
    gradient against fp64 autograd on the oracle for weights that are equal in both directions of an edge; weights that
    are not are refused in training (the backward reuses the forward operator as its transpose)"""
    from test_gpu_ops import rand_graph
    n, K, d = 400, 40, 64
    ei = rand_graph(n, 5, seed=3, self_loops=False, isolated=3)
    lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
    ew = ((lo * 7919 + hi * 104729) % 1000).float() / 1000 + 0.25          # w(i, j) == w(j, i)
    g = torch.Generator().manual_seed(2)
    x, w, b = torch.randn(n, K, generator=g), torch.randn(K, d, generator=g) / K ** 0.5, torch.randn(d, generator=g)
    dy = torch.randn(n, d, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    ref = torch.tanh(oracle.ops.gcn_conv(xr, ei, wr, br, edge_weight=ew.double()))
    ref.backward(dy.double())
    csr = ops.csr_from_edge_index(ei.to(device), n)
    with torch.no_grad():
        fwd = ops.gcn_conv(csr, x.to(device), w.to(device), b.to(device), ew.to(device), act="tanh")
    assert max_err(fwd, ref) < TOL
    xd, wd, bd = _leaf(x, device), _leaf(w, device), _leaf(b, device)
    got = ag.gcn_conv(xd, wd, bd, csr, act="tanh", edge_weight=ew.to(device))
    got.backward(dy.to(device))
    assert max_err(got, ref) < TOL
    for a, r in ((xd, xr), (wd, wr), (bd, br)):
        assert max_err(a.grad, r.grad) < TOL
    skew = ew.clone()
    skew[0] += 0.5
    with pytest.raises(ValueError):
        ag.gcn_conv(_leaf(x, device), wd, bd, csr, act="tanh", edge_weight=skew.to(device))
