"""GPU parity, module / model level: the drop-in nn.Modules built from the reference's
spec strings, run through libblgnn, against the oracle's interpreter of the same
spec strings and state_dict (fp32 max|delta| < 1e-5 with calibrated eval-BN)."""

import copy

import numpy as np
import pytest
import torch

import blgnn
import oracle
from blgnn.forward import bilevel_forward, get_initial_embd, all_molecule_pairs
from helpers import (TOL, max_err, build, layer_specs, state_dict_cpu, plain_batch, oracle_flags,
                     calibrate_bn)

pytestmark = pytest.mark.gpu


def _pairs(ds, B, seed=0):
    rng = np.random.default_rng(seed)
    idx = rng.choice(len(ds.pair_gids), size=min(B, len(ds.pair_gids)), replace=False)
    return ds.pair_gids[idx]


def _oracle_bilevel(model, flags, ds, specs, batch_gids, dtype=torch.float64):
    """lower sweep over all molecules -> init_x -> higher slice, on the CPU oracle"""
    of = oracle_flags(flags, ds)
    k = flags.last_lower_lyr_num
    allp = all_molecule_pairs(ds.gs_map.keys())
    lower_b = blgnn.BatchData(allp, ds, is_train=False, ignore_pairs=True)
    plain = plain_batch(lower_b, ds, torch.float32)
    sd32 = calibrate_bn(model, specs[:k + 1], plain, of)
    sd = {n: (v.to(dtype) if v.is_floating_point() else v) for n, v in sd32.items()}
    plain = plain_batch(lower_b, ds, dtype)
    oracle.run_layers(specs[:k + 1], sd, plain, of)
    init_x = plain["init_x"]
    # higher slice
    hb = blgnn.BatchData(batch_gids, ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(hb.batch_gids))
    ph = plain_batch(hb, ds, dtype, init_x=init_x, with_pairs=ids)
    # calibrate the upper BNs on the oracle's own init_x (fp32 weights)
    ph32 = plain_batch(hb, ds, torch.float32, init_x=init_x.float(), with_pairs=ids)
    sd32 = calibrate_bn(model, specs[k + 1:], ph32, of, first=k + 1)
    sd = {n: (v.to(dtype) if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs[k + 1:], sd, ph, of, first=k + 1)
    return init_x, acts


@pytest.mark.parametrize("lower,upper,d", [("gin", "gat", 64), ("gat", "gat", 64), ("gcn", "gcn", 64),
                                           ("gin", "gin", 128)])
def test_bignn_forward(device, lower, upper, d):
    ds = blgnn.make_synthetic(60, 400, seed=1, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type=lower,
                             higher_level_gnn_type=upper, D_lower=d, D_higher=d, device=device)
    model, data = build(flags, ds, seed=3)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 32)
    init_ref, acts_ref = _oracle_bilevel(model, flags, ds, specs, pairs)
    preds, bd = bilevel_forward(model, data, pairs)
    assert max_err(ds.interaction_combo_nxgraph.init_x, init_ref) < TOL
    assert max_err(preds, acts_ref[-1]) < TOL
    # every intermediate activation of the higher slice as well
    for got, ref in zip(model.acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < TOL


@pytest.mark.parametrize("transform", ["f16x3", "bf16x6", "f32"])
def test_bignn_forward_under_each_feature_transform(device, transform):
    """the three ways the fused GIN layer runs its dense transforms (fp16 two-piece split, bf16 three-piece
    split, exact-f32 MFMA) all meet the same 1e-5 bar on the whole bi-level forward at d=128"""
    ds = blgnn.make_synthetic(90, 500, seed=4, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                             feature_transform=transform)
    model, data = build(flags, ds, seed=9)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 48)
    init_ref, acts_ref = _oracle_bilevel(model, flags, ds, specs, pairs)
    preds, bd = bilevel_forward(model, data, pairs)
    assert max_err(ds.interaction_combo_nxgraph.init_x, init_ref) < TOL
    assert max_err(preds, acts_ref[-1]) < TOL


@pytest.mark.parametrize("group", ["src", "dst"])
def test_decagon_forward(device, group):
    ds = blgnn.make_synthetic(80, 600, seed=2, device=device)
    flags = blgnn.make_flags("higher_level_gnn", device=device, gat_softmax_group=group)
    model, data = build(flags, ds, seed=5)
    ds.init_interaction_graph_embds(device)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 64)
    bd = blgnn.BatchData(pairs, ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(ds.gs_map.get)(bd.batch_gids))
    init_x = ds.interaction_combo_nxgraph.init_x
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs, plain_batch(bd, ds, torch.float32, init_x=init_x, with_pairs=ids), of)
    sd = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, init_x=init_x, with_pairs=ids), of)
    model.use_layers = 'all'
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, ref in zip(model.acts[1:], acts[1:]):
        assert max_err(got, ref) < TOL


def test_llgnn_forward(device):
    ds = blgnn.make_synthetic(50, 300, seed=4, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gin", device=device)
    model, data = build(flags, ds, seed=7)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 24)
    bd = blgnn.BatchData(pairs, ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs, plain_batch(bd, ds, torch.float32, with_pairs=ids), of)
    sd = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, ref in zip(model.acts[1:], acts[1:]):
        assert max_err(got, ref) < TOL


def test_mhcaddi_gmn_forward(device):
    ds = blgnn.make_synthetic(40, 200, seed=6, device=device)
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=64, device=device)
    model, data = build(flags, ds, seed=9)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 16)
    bd = blgnn.BatchData(pairs, ds, is_train=False, unique_graphs=False)
    ids = torch.arange(2 * len(pairs)).reshape(-1, 2)
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs, plain_batch(bd, ds, torch.float32, with_pairs=ids), of)
    sd = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, ref in zip(model.acts[1:], acts[1:]):
        assert max_err(got, ref) < TOL


def test_gmn_gru_node_update_forward(device):
    """GMNPropagator(f_node='GRU') (model/layers_gmn.py:27-30,48-50, non-default): GRUCell(cat(m_sum, u_sum), x)"""
    ds = blgnn.make_synthetic(40, 200, seed=6, device=device)
    flags = blgnn.make_flags("lower_level_gmn", lower_level_gnn_type="gin", D_lower=64, device=device, gmn_f_node="GRU")
    model, data = build(flags, ds, seed=9)
    assert isinstance(model.layers[1].f_node, torch.nn.GRUCell)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 16)
    bd = blgnn.BatchData(pairs, ds, is_train=False, unique_graphs=False)
    ids = torch.arange(2 * len(pairs)).reshape(-1, 2)
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs, plain_batch(bd, ds, torch.float32, with_pairs=ids), of)
    sd = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, ref in zip(model.acts[1:], acts[1:]):
        assert max_err(got, ref) < TOL


@pytest.mark.parametrize("node_model", ["gat", "gcn"])
def test_drugcombo_meta_forward(device, node_model):
    ds = blgnn.make_synthetic(70, 500, seed=8, device=device, num_edge_types=2, name="drugcombo_syn")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", dataset="drugcombo", lower_level_gnn_type="gin",
                             device=device, multi_edge_node_model=node_model)
    model, data = build(flags, ds, seed=11, num_hyper_edge_feat=3)
    specs = layer_specs(flags)
    pairs = _pairs(ds, 32)
    init_ref, acts_ref = _oracle_bilevel(model, flags, ds, specs, pairs)
    preds, bd = bilevel_forward(model, data, pairs)
    assert preds.shape[1] == 3
    assert max_err(ds.interaction_combo_nxgraph.init_x, init_ref) < TOL
    for got, ref in zip(model.acts[1:], acts_ref[1:]):
        assert max_err(got, ref) < TOL


def test_block_diagonal_independence_and_permutation(device):
    """a molecule's lower-level output does not depend on what else is in the batch,
    and permuting the batch order permutes the pooled rows (size-independent properties,
    checked at the DrugBank-shape size: 1 326 molecules)."""
    ds = blgnn.make_synthetic(1326, 3000, seed=10, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gin", D_lower=128, device=device)
    model, data = build(flags, ds, seed=13)
    model.use_layers = 'lower_layers'
    gids = np.asarray(ds.molecules.gids)
    full = model(blgnn.BatchData(all_molecule_pairs(gids), ds, is_train=False, ignore_pairs=True)).clone()
    perm = np.random.default_rng(0).permutation(len(gids))
    pg = gids[perm]
    permuted = model(blgnn.BatchData(all_molecule_pairs(pg), ds, is_train=False, ignore_pairs=True))
    # not bit-identical: a molecule that straddles a 64-row tile edge sums its out-of-tile
    # neighbours after the in-tile ones, and tile edges move with the batch composition
    assert max_err(permuted, full[torch.as_tensor(perm, device=device)]) < 1e-6
    sub = gids[100:140]
    part = model(blgnn.BatchData(all_molecule_pairs(sub), ds, is_train=False, ignore_pairs=True))
    assert max_err(part, full[100:140]) < 1e-6


def test_engine_graph_replay_and_two_stream_overlap_match_eager(device):
    """blgnn/engine.py: HIP-graph replay and the two-stream software pipeline (double-buffered
    init_x) give exactly the eager results, step after step."""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(200, 1500, seed=21, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=64, D_higher=64, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=5)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 64), device=device)
    ds.init = True
    rng = np.random.default_rng(3)
    uppers = [blgnn.BatchData(ds.pair_gids[rng.choice(len(ds.pair_gids), 96, replace=False)], ds, is_train=False)
              for _ in range(3)]
    plain_engine = BiLevelEngine(model, lower, uppers, g, use_graph=False, fuse_lower=False)
    plain = [plain_engine.eager_step(i).clone() for i in range(3)]
    init_plain = g.init_x.clone()
    ref_engine = BiLevelEngine(model, lower, uppers, g, use_graph=False, fuse_lower=True)
    assert ref_engine.fused is not None and ref_engine.fused.large is not None      # both paths exercised
    refs = [ref_engine.eager_step(i).clone() for i in range(3)]
    # one-launch lower stack (whole-molecule tiles) == layer-by-layer kernels, to rounding
    assert max_err(g.init_x, init_plain) < TOL
    for a, b in zip(refs, plain):
        assert max_err(a, b) < TOL
    # The sweep's input changes from step to step (written on the sweep's own stream, as a sampler that
    # refreshes features would), a long blocker delays every interaction-level pass, and EVERY step's
    # scores are compared with the eager forward of the same inputs: a missing event wait or a wrong
    # parity buffer (step k+2's sweep overwriting init_x while step k's upper pass still reads it) fails.
    scales = [1.0 + 0.25 * k for k in range(7)]
    fresh = ds.pair_gids[rng.choice(len(ds.pair_gids), 96, replace=False)]      # a batch outside the list
    def which(k):
        return fresh if k == 5 else k % 3
    blk_a, blk_b = torch.randn(2048, 2048, device=device), torch.randn(2048, 2048, device=device)

    def live_inputs(engine):
        """the feature tensors the engine's sweep reads (the fused stack keeps its own tiled copies)"""
        if engine.fused is None:
            xs = [lower.merge_data['merge'].x]
        else:
            xs = [b.merge_data['merge'].x for b in (engine.fused.small, engine.fused.large) if b is not None]
        if not hasattr(engine, "_x0"):
            engine._x0 = [x.clone() for x in xs]
        return list(zip(xs, engine._x0))

    def set_inputs(engine, scale):
        for x, x0 in live_inputs(engine):
            x.copy_(x0 * scale)

    def eager_refs(engine):
        out = []
        for k in range(7):
            set_inputs(engine, scales[k])
            out.append(engine.eager_step(which(k)).clone())
        set_inputs(engine, 1.0)
        torch.cuda.synchronize()
        return out

    def pipelined(engine):
        engine.capture()
        got = []
        for k in range(7):
            sl = engine.stream_l if engine.overlap else torch.cuda.current_stream()
            su = engine.stream_u if engine.overlap else torch.cuda.current_stream()
            with torch.cuda.stream(sl):
                set_inputs(engine, scales[k])
            with torch.cuda.stream(su):
                for _ in range(4):
                    torch.mm(blk_a, blk_b)
            o = engine.step(which(k))
            with torch.cuda.stream(su):
                got.append(o.clone())                   # the score buffer is reused two steps later
        engine.synchronize()
        set_inputs(engine, 1.0)
        torch.cuda.synchronize()
        return got

    refs_f = eager_refs(ref_engine)
    for overlap in (False, True):
        with BiLevelEngine(model, lower, uppers, g, use_graph=True, overlap=overlap, fuse_lower=True) as eng:
            for k, o in enumerate(pipelined(eng)):
                assert torch.equal(o, refs_f[k]), (overlap, k)
    # per-layer kernels, two streams, with the readout riding in the interaction-level graph
    # and at the tail of the sweep's graph: the same numbers as the eager per-layer forward
    refs_p = eager_refs(plain_engine)
    for defer, cus in ((True, None), (False, None), (False, 20), (True, 12)):
        # cus: each stream on its own compute units of every XCD (blgnn/streams.py)
        eng = BiLevelEngine(model, lower, uppers, g, use_graph=True, overlap=True, fuse_lower=False,
                            readout_with_upper=defer, lower_cus_per_xcd=cus)
        assert eng.readout_with_upper == defer
        if cus and torch.cuda.get_device_properties(device).multi_processor_count == 256:
            assert eng.lower_cus_per_xcd == cus
        for k, o in enumerate(pipelined(eng)):
            assert torch.equal(o, refs_p[k]), (defer, cus, k)
        eng.close()
        eng.close()                                     # idempotent
        with pytest.raises(RuntimeError):
            eng.step(0)


def test_engine_parities_on_unrelated_streams_match_eager(device):
    """The engine's double buffering must not lean on stream order: by default both parities' sweeps share one stream
    and both interaction passes another, which would serialise -- and hide -- any buffer the two parities' graphs both
    touch.  Here each parity replays its graphs on its OWN stream pair (``parity_streams``), ordered by nothing but the
    events of ``step()``; a long blocker holds back one parity at a time so that the other parity's sweep AND
    interaction pass run to completion beside it, and every step's scores must be the eager forward's, bit for bit.
    (Round 3's twin-stream experiment produced wrong scores in a harness that also REWROTE the sweep's input features
    per step on the sweep's stream: the merged atom features are the one buffer both parities' sweeps read, and nothing
    but the shared stream ordered that rewrite against the other parity's sweep.  The engine itself never writes them:
    second half of this test.)"""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(200, 1500, seed=27, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=64, D_higher=64, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=6)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 64), device=device)
    ds.init = True
    rng = np.random.default_rng(4)
    batches = [ds.pair_gids[rng.choice(len(ds.pair_gids), 96, replace=False)] for _ in range(8)]
    template = blgnn.BatchData(batches[0], ds, is_train=False)
    plain = BiLevelEngine(model, lower, template, g, use_graph=False)
    refs = [plain.eager_step(b).clone() for b in batches]
    blk_a, blk_b = torch.randn(4096, 4096, device=device), torch.randn(4096, 4096, device=device)
    for fused_readout in (False, True):
        flags.fuse_readout = fused_readout
        blgnn.set_flags(flags)
        with BiLevelEngine(model, lower, template, g, use_graph=True, overlap=True, parity_streams=True) as eng:
            assert eng.parity_streams and eng._sl[0] is not eng._sl[1] and eng._su[0] is not eng._su[1]
            eng.capture()
            got = []
            for k, b in enumerate(batches):
                p = k & 1
                # hold back this parity (its sweep on even rounds, its interaction pass on odd ones): the next step, of the
                # other parity, overtakes it completely
                with torch.cuda.stream(eng._sl[p] if (k // 2) % 2 == 0 else eng._su[p]):
                    for _ in range(6):
                        torch.mm(blk_a, blk_b)
                o = eng.step(b)
                with torch.cuda.stream(eng._su[p]):
                    got.append(o.clone())                   # the score buffer is reused two steps later
            eng.synchronize()
            for k, (o, r) in enumerate(zip(got, refs)):
                if fused_readout:                           # float atomics in the pooled rows: rounding, not bits
                    assert max_err(o, r) < 1e-6, k
                else:
                    assert torch.equal(o, r), k


def test_engine_exchange_after_first_transform_is_bitwise_the_replicated_forward(device):
    """multi-GPU decomposition, run at one rank: the first interaction layer's row-wise half (h = pooled @ W and
    the two attention scores) applied to the rank's own pooled rows, the [N, d + 2] table exchanged instead of
    [N, 5 d], the graph half on the gathered table -- the same bits as the replicated layer"""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(150, 1100, seed=23, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=7)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    ds.init = True
    uppers = [blgnn.BatchData(_pairs(ds, 80, seed=k), ds, is_train=False) for k in range(2)]
    plain = BiLevelEngine(model, lower, uppers, g, use_graph=False)
    refs = [plain.eager_step(i).clone() for i in range(2)]
    for use_graph in (False, True):
        with BiLevelEngine(model, lower, uppers, g, use_graph=use_graph, exchange_transformed=True) as eng:
            assert eng.exchange_transformed
            eng.capture()
            for i in range(4):
                out = eng.step(i)
                eng.synchronize()
                assert torch.equal(out, refs[i % 2]), (use_graph, i)


def test_engine_last_interaction_layer_pruned_to_scored_rows(device):
    """a rank of a sharded job scores few pairs out of all the interaction rows: the LAST interaction-level GAT layer
    then computes only the rows the pair scorer reads (blgnn_gat_aggregate's row list; model/layers_link_pred.py:56-61
    reads nothing else).  Same kernel on those rows: the scores are bit for bit the unpruned ones; and the op itself
    leaves every other row untouched"""
    from blgnn import ops
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(400, 9000, seed=29, device=device)      # mean interaction degree ~45: the wide gather
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=9)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    ds.init = True
    uppers = [blgnn.BatchData(_pairs(ds, 48, seed=k), ds, is_train=False) for k in range(2)]
    plain = BiLevelEngine(model, lower, uppers, g, use_graph=False, prune_last=False)
    assert plain._prune_layer() is not None and not plain.prune_last
    refs = [plain.eager_step(i).clone() for i in range(2)]
    for use_graph in (False, True):
        with BiLevelEngine(model, lower, uppers, g, use_graph=use_graph, prune_last=True) as eng:
            eng.capture()
            for i in range(4):
                out = eng.step(i)
                eng.synchronize()
                assert torch.equal(out, refs[i % 2]), (use_graph, i)
    # op level: listed rows (with a duplicate) equal the full result, the others keep the sentinel
    csr = uppers[0].merge_higher_level['merge'].csr()
    n, d = csr.n, 128
    gen = torch.Generator().manual_seed(3)
    h = torch.randn(n, d, generator=gen).to(device)
    sc = torch.randn(2, n, generator=gen).to(device)
    bias = torch.randn(d, generator=gen).to(device)
    full = ops.gat_from_transform(csr, h, sc, bias, act="relu")
    rows = torch.tensor([5, 17, 5, n - 1, 0, 123], dtype=torch.int64, device=device)
    part = torch.full_like(full, -7.0)
    ops.gat_from_transform(csr, h, sc, bias, act="relu", out=part, rows=rows)
    keep = torch.zeros(n, dtype=torch.bool, device=device)
    keep[rows] = True
    assert torch.equal(part[keep], full[keep]) and bool((part[~keep] == -7.0).all())
    # negative entries (the padding of the engine's de-duplicated list) are skipped
    rows = torch.tensor([-1, 5, -1, -1, 17, n - 1, -1], dtype=torch.int64, device=device)
    part = torch.full_like(full, -7.0)
    ops.gat_from_transform(csr, h, sc, bias, act="relu", out=part, rows=rows)
    keep = torch.zeros(n, dtype=torch.bool, device=device)
    keep[rows[rows >= 0]] = True
    assert torch.equal(part[keep], full[keep]) and bool((part[~keep] == -7.0).all())


def test_engine_follows_weight_updates_after_capture(device):
    """captured graphs hold pointers to tensors derived from the weights (packed images, folded BN affines):
    an in-place weight / running-stat update after capture() must not leave step() replaying stale values"""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(120, 900, seed=22, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=6)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    ds.init = True
    upper = blgnn.BatchData(_pairs(ds, 64), ds, is_train=False)
    with BiLevelEngine(model, lower, upper, g, use_graph=True, overlap=True) as eng:
        eng.capture()
        before = eng.step(0).clone()
        eng.synchronize()
        with torch.no_grad():
            model.layers[0].conv.nn[0].weight.mul_(0.5)
            model.layers[1].bn.running_var.add_(0.3)
            model.layers[7].conv.att.mul_(-1.0)
        after = eng.step(0)
        eng.synchronize()
        after = after.clone()
        ref = eng.eager_step(0)
        torch.cuda.synchronize()
        assert torch.equal(after, ref)
        assert not torch.equal(after, before)


@pytest.mark.parametrize("node_aggr", ["avg_pool", "deepsets", "gmn_aggr"])
def test_readout_styles_forward(device, node_aggr):
    """single-scale mean, DeepSets and the gated GMN aggregator (model/layers_aggregation.py:45-95)"""
    ds = blgnn.make_synthetic(40, 200, seed=12, device=device)
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gin", D_lower=64, device=device,
                             node_aggr=node_aggr)
    model, data = build(flags, ds, seed=4)
    specs = layer_specs(flags)
    bd = blgnn.BatchData(_pairs(ds, 20), ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    of = oracle_flags(flags, ds)
    sd32 = calibrate_bn(model, specs, plain_batch(bd, ds, torch.float32, with_pairs=ids), of)
    sd = {n: (v.double() if v.is_floating_point() else v) for n, v in sd32.items()}
    acts = oracle.run_layers(specs, sd, plain_batch(bd, ds, torch.float64, with_pairs=ids), of)
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, ref in zip(model.acts[1:], acts[1:]):
        assert max_err(got, ref) < TOL


def test_fused_readout_is_the_default_and_matches_the_separate_launch(device):
    """north_star "with the node readout fused in": by default the lower GIN layers accumulate the multi-scale
    readout from their epilogues (no segment-reduce launch); == the separate launch to fp32 rounding, direct
    into init_x, and the engine's captured graphs replay it"""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(120, 600, seed=6, device=device)
    outs = {}
    for fuse in (True, False):
        flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                                 higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                                 fuse_readout=fuse)
        model, data = build(flags, ds, seed=7)
        calls = []
        orig = blgnn.ops.segment_reduce_multi
        blgnn.ops.segment_reduce_multi = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            preds, bd = bilevel_forward(model, data, _pairs(ds, 64))
        finally:
            blgnn.ops.segment_reduce_multi = orig
        assert (len(calls) == 0) == fuse
        outs[fuse] = (preds.clone(), ds.interaction_combo_nxgraph.init_x.clone())
    assert max_err(outs[True][1], outs[False][1]) < 1e-6
    assert max_err(outs[True][0], outs[False][0]) < 1e-6
    # engine: eager and captured replays of the fused-readout sweep agree to rounding (atomics may reorder)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", higher_level_gnn_type="gat",
                             D_lower=128, D_higher=128, device=device, fuse_readout=True)
    model, data = build(flags, ds, seed=7)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    upper = blgnn.BatchData(_pairs(ds, 64), ds, is_train=False)
    with BiLevelEngine(model, lower, upper, g, use_graph=False) as e0:
        ref = e0.eager_step(0).clone()
    with BiLevelEngine(model, lower, upper, g, use_graph=True, overlap=True) as eng:
        eng.capture()
        for _ in range(3):
            out = eng.step(0)
        eng.synchronize()
        assert max_err(out, ref) < 1e-6


def test_engine_target_row_sharded_interaction_level_one_rank(device):
    """``BiLevelEngine(shard_upper=True)`` at one rank (own rows = every row, no collective): each interaction layer's graph
    half through the row list, the next layer's row-wise half on those rows, three captured pieces per pass -- the
    same kernels on the same rows as the replicated pass: bit for bit its scores (the last layer pruned or not)"""
    from blgnn.engine import BiLevelEngine
    ds = blgnn.make_synthetic(400, 9000, seed=29, device=device)
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin",
                             higher_level_gnn_type="gat", D_lower=128, D_higher=128, device=device,
                             fuse_readout=False)   # bitwise comparisons: no float atomics in the readout
    model, data = build(flags, ds, seed=9)
    lower = blgnn.BatchData(all_molecule_pairs(ds.molecules.gids), ds, is_train=False, ignore_pairs=True)
    g = ds.interaction_combo_nxgraph
    g.init_x = torch.zeros((ds.num_graphs(), 5 * 128), device=device)
    ds.init = True
    uppers = [blgnn.BatchData(_pairs(ds, 48, seed=k), ds, is_train=False) for k in range(2)]
    plain = BiLevelEngine(model, lower, uppers, g, use_graph=False, prune_last=False)
    refs = [plain.eager_step(i).clone() for i in range(2)]
    for use_graph in (False, True):
        for prune in (False, True):
            with BiLevelEngine(model, lower, uppers, g, use_graph=use_graph, exchange_transformed=True,
                               shard_upper=True, prune_last=prune) as eng:
                assert eng.shard_upper and eng.n_upper_segments == 3
                eng.capture()
                for i in range(4):
                    out = eng.step(i)
                    eng.synchronize()
                    assert torch.equal(out, refs[i % 2]), (use_graph, prune, i)


def _run_shard_workers(world, backend):
    import os
    import socket
    import subprocess
    import sys
    import conftest
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "BLGNN_WORKER_BACKEND": backend})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(conftest.ROOT, "tests", "shard_upper_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=400)[0])
    finally:
        for p in procs:                                   # exact PIDs started here
            if p.poll() is None:
                p.kill()
    if backend == "nccl" and any(p.returncode == 77 for p in procs):
        pytest.skip("RCCL did not initialise on this box: " + outs[0][-300:])
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d:\n%s" % (r, outs[r][-3000:])
    return outs


def test_engine_collectives_over_rccl_one_rank(device):
    """the engine's all-gathers as REAL RCCL calls (one rank, backend nccl, ``BiLevelEngine(collective=True)``): the
    collective of the sweep and the two of the target-row-sharded interaction pass, issued between replays of the captured
    graphs on the engine's two streams, eager and captured, sharded and replicated -- bit for bit the scores of an engine
    without any collective.  (Two ranks cannot share one GPU under RCCL: the two-rank test below runs over gloo.)"""
    out = _run_shard_workers(1, "nccl")[0]
    assert "nccl, world 1" in out


def test_engine_target_row_sharded_interaction_level_two_ranks(device):
    """the same with TWO ranks sharing the GPU over gloo (tests/shard_upper_worker.py): a rank computes its own target
    rows of every interaction layer but the last, all-gathers the next layer's transformed rows, and scores -- equal to
    the replicated pass and to a one-rank engine over all drugs"""
    _run_shard_workers(2, "gloo")
