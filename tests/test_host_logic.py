"""CPU: host-side pieces of the path -- merge, batch building, negative sampling, the
sampler, synthetic generator statistics, gexf loading -- against the oracle / reference rules."""

import numpy as np
import pytest
import torch

import blgnn
import oracle
import fake_backend
from blgnn.batch import BatchData, ordered_unique
from blgnn.config import set_flags
from blgnn.graph import GraphData, merge_from_data_list, to_undirected_host
from blgnn.sampler import RandomSampler
from helpers import build, layer_specs, max_err


def test_to_undirected_host_matches_oracle():
    rng = np.random.default_rng(0)
    e = rng.integers(0, 30, size=(200, 2))
    got = to_undirected_host(e, 30)
    ref = oracle.to_undirected(torch.as_tensor(e.T.copy()), 30).numpy()
    assert np.array_equal(got, ref)
    assert to_undirected_host(np.zeros((0, 2)), 5).shape == (2, 0)


def test_merge_from_data_list_and_store_merge_match_oracle():
    ds = blgnn.make_synthetic(12, 30, seed=3, device="cpu")
    gids = [5, 2, 9, 2]                                         # duplicates allowed in pair mode
    singles = [ds.molecules.single(g) for g in gids]
    ref = oracle.merge_graphs([(s.x, s.edge_index) for s in singles])
    for merged in (merge_from_data_list(singles, unique_graphs=False), ds.molecules.merge(gids, unique_graphs=False)):
        m = merged["merge"]
        assert torch.equal(m.x, ref["x"]) and torch.equal(m.edge_index, ref["edge_index"])
        assert torch.equal(m.batch, ref["batch"]) and merged["ind_list"] == ref["ind_list"]
        assert merged["edge_ind_list"] == ref["edge_ind_list"] and merged["gids_to_batch_ind"] is None
        assert m._seg.tolist() == [0] + [e for _, e in ref["ind_list"]]
        assert m.is_undirected()
    uniq = merge_from_data_list({g: ds.molecules.single(g) for g in [5, 2, 9]}, unique_graphs=True)
    assert uniq["gids_to_batch_ind"] == {5: 0, 2: 1, 9: 2} and uniq["graph_sizes"].tolist() == \
        [int(ds.molecules.sizes[ds.molecules.gid_to_pos[g]]) for g in [5, 2, 9]]


def test_batchdata_unique_order_and_fields():
    ds = blgnn.make_synthetic(20, 60, seed=4, device="cpu")
    set_flags(blgnn.make_flags("lower_level_gnn", device="cpu"))
    pairs = np.asarray([[7, 3], [3, 9], [7, 1]])
    assert ordered_unique(pairs).tolist() == [7, 3, 9, 1]        # dict insertion order of src/batch.py:131-136
    bd = BatchData(pairs, ds, is_train=False, ignore_pairs=True)
    assert bd._graph_gids == [7, 3, 9, 1] and bd.merge_data["gids_to_batch_ind"] == {7: 0, 3: 1, 9: 2, 1: 3}
    assert bd.batch_interaction_inds == [7, 3, 3, 9, 7, 1] and bd.pair_list == []
    pair_mode = BatchData(pairs, ds, is_train=False, ignore_pairs=True, unique_graphs=False)
    assert pair_mode._graph_gids == [7, 3, 3, 9, 7, 1] and pair_mode.merge_data["merge"].num_graphs == 6
    with pytest.raises(ValueError):                               # unknown pair, as look_up_pair_by_gids does
        known = {tuple(sorted(p)) for p in ds.pair_gids.tolist()}
        a, b = next((i, j) for i in range(20) for j in range(i + 1, 20) if (i, j) not in known)
        BatchData(np.asarray([[a, b]]), ds, is_train=False)


def test_shared_molecules_of_a_pair_batch():
    """pair batch (src/config.py:258: batch_unique_graphs=False) -> the distinct molecules merged once in order of
    first appearance + the row map that reproduces every copy's rows, edges and graph ids"""
    ds = blgnn.make_synthetic(20, 60, seed=4, device="cpu")
    set_flags(blgnn.make_flags("lower_level_gnn", device="cpu"))
    pairs = np.asarray([[7, 3], [3, 9], [7, 1], [9, 7]])
    bd = BatchData(pairs, ds, is_train=False, ignore_pairs=True, unique_graphs=False)
    sm = bd.shared_molecules()
    assert sm is bd.shared_molecules()                            # built once per batch
    m, u = bd.merge_data["merge"], sm.graph
    assert u.num_graphs == 4 and sm.row_map.dtype == torch.int32 and sm.row_map.shape[0] == m.x.shape[0]
    rm = sm.row_map.long()
    assert torch.equal(u.x.index_select(0, rm), m.x)
    # graph ids: copy k of the batch is distinct molecule [7, 3, 9, 1].index(gid_k)
    first = {7: 0, 3: 1, 9: 2, 1: 3}
    want = torch.as_tensor([first[g] for g in bd._graph_gids])[m.batch]
    assert torch.equal(u.batch[rm], want)
    # edges: mapping both ends of every merged edge lands on an edge of the distinct graph
    key = lambda ei, n: set((ei[0] * n + ei[1]).tolist())
    assert key(rm[m.edge_index], u.num_nodes) == key(u.edge_index, u.num_nodes)
    assert BatchData(pairs, ds, is_train=False, ignore_pairs=True).shared_molecules() is None    # unique-graph batch


def test_negative_sampling_rules():
    ds = blgnn.make_synthetic(40, 150, seed=6, device="cpu")
    flags = blgnn.make_flags("lower_level_gnn", device="cpu")
    flags.enforce_sampling_amongst_same_graphs = True
    set_flags(flags)
    pos = ds.pair_gids[:16]
    bd = BatchData(pos, ds, sampled_gids=np.unique(pos), is_train=True, rng=np.random.default_rng(1))
    neg = bd.negative_pair_gids
    assert len(neg) == 16 and len(bd.batch_gids) == 32 and bd.true_labels().tolist() == [1] * 16 + [0] * 16
    known = {tuple(sorted(p)) for p in ds.pair_gids.tolist()}
    sampled = set(np.unique(pos).tolist())
    seen = set()
    for a, b in neg.tolist():
        assert a != b and tuple(sorted((a, b))) not in known        # never a positive edge, never a self pair
        assert a in sampled and b in sampled                          # among the batch's own graphs
        assert tuple(sorted((a, b))) not in seen
        seen.add(tuple(sorted((a, b))))
    assert bd.pair_list[20].true_label == 0 and bd.pair_list[3].true_label == 1


def _negatives_one_candidate_at_a_time(ds, pos_pairs, sampled, rng, enforce=True):
    """the acceptance loop of src/batch.py:75-98 as the reference writes it (one candidate per python iteration): the
    whole-array sampler of blgnn/batch.py must pick the same negatives from the same random draws"""
    gs_map = ds.gs_map
    want = len(pos_pairs)
    n = len(sampled)
    want = int(min(want, n * (n - 1) // 2 - len(pos_pairs)))
    N = np.int64(ds.num_graphs())
    rows = np.vectorize(gs_map.get)
    bp = rows(pos_pairs)
    pos = np.union1d(ds.positive_pair_keys(), np.minimum(bp[:, 0], bp[:, 1]) * N + np.maximum(bp[:, 0], bp[:, 1]))
    chosen, seen, cursor = [], set(), 0
    while len(chosen) < want:
        m = max(2 * (want - len(chosen)), 16)
        a = sampled[(cursor + np.arange(m)) % n]
        cursor += m
        b = rng.choice(sampled, size=m)
        ra, rb = rows(a), rows(b)
        key = np.minimum(ra, rb) * N + np.maximum(ra, rb)
        ok = (a != b) & ~np.isin(key, pos)
        for x, y, k, good in zip(a, b, key, ok):
            if good and int(k) not in seen:
                seen.add(int(k))
                chosen.append((int(x), int(y)))
                if len(chosen) == want:
                    break
    return np.asarray(chosen)


@pytest.mark.parametrize("seed,batch", [(1, 16), (2, 64), (3, 200)])
def test_vectorised_negative_sampler_picks_what_the_candidate_loop_picks(seed, batch):
    ds = blgnn.make_synthetic(60, 400, seed=10 + seed, device="cpu")
    set_flags(blgnn.make_flags("lower_level_gnn", device="cpu"))
    pos = ds.pair_gids[np.random.default_rng(seed).choice(len(ds.pair_gids), size=batch, replace=False)]
    sampled = np.unique(pos)
    bd = BatchData(pos, ds, sampled_gids=sampled, is_train=True, rng=np.random.default_rng(100 + seed))
    ref = _negatives_one_candidate_at_a_time(ds, pos, sampled, np.random.default_rng(100 + seed))
    assert np.array_equal(bd.negative_pair_gids, ref)
    assert len(bd.pair_list) == len(bd.batch_gids) and bd.pair_list[len(pos)].true_label == 0


def test_random_sampler_covers_every_pair_once_per_epoch():
    ds = blgnn.make_synthetic(15, 50, seed=2, device="cpu")
    data = blgnn.TrainData(ds, device="cpu")
    s = RandomSampler(data, batch_size=16, seed=0)
    got = []
    for _ in range(4):                                           # 50 pairs -> 16+16+16+2
        b, uniq, sub = s.sample_next_training_batch()
        assert sub is None and set(uniq.tolist()) == set(b.reshape(-1).tolist())
        got += [tuple(p) for p in b.tolist()]
    assert len(got) == 50 and set(got) == {tuple(p) for p in ds.pair_gids.tolist()}
    assert len(s.sample_next_training_batch()[0]) == 16          # next epoch starts


def test_synthetic_generator_matches_drugbank_statistics():
    ds = blgnn.make_synthetic(1326, 29240, seed=20200616, device="cpu")
    m = ds.molecules
    assert m.num_node_feat == 52 and bool((m.x.sum(1) == 6).all())      # F_in = 52, six one-hots per atom
    assert 22 <= m.sizes.mean() <= 32 and m.sizes.min() >= 2 and m.sizes.max() <= 457
    assert 1.9 <= m.esizes.sum() / m.sizes.sum() <= 2.3                 # nnz/atom ~ 2.13
    deg = np.bincount(ds.interaction_combo_nxgraph.edges.reshape(-1), minlength=1326)
    assert len(ds.pair_gids) == 29240 and deg.max() > 4 * deg.mean()    # power-law skew
    again = blgnn.make_synthetic(1326, 29240, seed=20200616, device="cpu")
    assert torch.equal(again.molecules.x, m.x) and np.array_equal(again.pair_gids, ds.pair_gids)


def test_get_initial_embd_chunked_equals_single_sweep(monkeypatch):
    fake_backend.install(monkeypatch)
    from blgnn.forward import get_initial_embd
    ds = blgnn.make_synthetic(21, 60, seed=8, device="cpu")
    flags = blgnn.make_flags("lower_level_gnn_higher_level", lower_level_gnn_type="gin", D_lower=16,
                             D_higher=16, device="cpu")
    model, data = build(flags, ds, seed=1)
    whole = get_initial_embd(data, model).clone()
    init_whole = ds.interaction_combo_nxgraph.init_x.clone()
    chunked = get_initial_embd(data, model, chunk_pairs=3)
    assert whole.shape == (21, 80) and max_err(chunked, whole) < 1e-6
    assert max_err(ds.interaction_combo_nxgraph.init_x, init_whole) < 1e-6


@pytest.mark.parametrize("node_aggr", ["avg_pool", "deepsets", "gmn_aggr"])
def test_readout_styles_wiring(monkeypatch, node_aggr):
    """non-default readouts of model/layers_aggregation.py:45-56,78-95 through the factory"""
    fake_backend.install(monkeypatch)
    ds = blgnn.make_synthetic(18, 50, seed=9, device="cpu")
    flags = blgnn.make_flags("lower_level_gnn", lower_level_gnn_type="gcn", D_lower=16, device="cpu",
                             node_aggr=node_aggr)
    model, data = build(flags, ds, seed=2)
    specs = layer_specs(flags)
    bd = BatchData(ds.pair_gids[:9], ds, is_train=False)
    ids = torch.as_tensor(np.vectorize(bd.merge_data['gids_to_batch_ind'].get)(bd.batch_gids))
    from helpers import state_dict_cpu, plain_batch, oracle_flags
    ref = oracle.run_layers(specs, state_dict_cpu(model), plain_batch(bd, ds, with_pairs=ids), oracle_flags(flags, ds))
    model.acts = [bd.merge_data['merge'].x]
    for layer in model.layers[:-1]:
        model.acts.append(layer(model.acts[-1], bd, model))
    for got, r in zip(model.acts[1:], ref[1:]):
        assert max_err(got, r) < 1e-5


def test_engine_scored_rows_are_listed_once():
    """BiLevelEngine.dedup_rows (the row list of the pruned last interaction layer): every distinct id once, ascending,
    repeats replaced by -1 -- the gather kernels skip negative entries -- and the shape of the input kept (capturable)"""
    from blgnn.engine import BiLevelEngine
    rng = np.random.default_rng(0)
    for m, hi in ((1, 5), (64, 7), (4096, 300), (4096, 10 ** 6)):
        ids = torch.as_tensor(rng.integers(0, hi, size=m))
        out = BiLevelEngine.dedup_rows(ids)
        assert out.shape == ids.shape and out.dtype == torch.int64
        kept = out[out >= 0]
        assert torch.equal(kept, torch.unique(ids))                   # distinct, ascending
        assert int((out == -1).sum()) == m - kept.numel()
