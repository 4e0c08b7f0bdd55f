"""Dataset objects the hot path reads, and the deterministic synthetic generator.

``BiGNNDataset`` carries exactly what the layers and ``BatchData`` consume from the
reference's utils/data/dataset.py: ``gs_map`` / ``id_map`` (:56-57), the interaction
graph(s) built from the pair list (:72-113), ``num_labels``, the ``init_x`` policies
(:166-210).  Molecules live pre-tensorised in a ``MoleculeStore``.

``make_synthetic`` follows SURVEY section 8(d): DrugBank-shaped molecules (log-normal
sizes, spanning tree + ring closures -> nnz/atom ~ 2.13, six categorical features
one-hot encoded to F_in = 52) and a power-law (Chung-Lu) interaction graph.
"""

import math

import numpy as np
import torch

from .graph import MoleculeStore, InteractionGraph

__all__ = ["BiGNNDataset", "TrainData", "make_synthetic", "one_hot_from_codes", "FEATURE_CARDS"]

FEATURE_CARDS = (32, 2, 2, 2, 7, 7)     # atom_type, aromatic, acceptor, donor, hybridization, num_h
DRUGCOMBO_CARDS = (20, 2, 2, 2, 7, 7)   # F_in = 40 (DrugCombo one-hot width)


class BiGNNDataset(object):
    def __init__(self, name, molecules, pair_gids, pair_labels, multi_edge_types=False):
        self.name = name
        self.molecules = molecules
        self.gs_map = {gid: i for i, gid in enumerate(molecules.gids)}
        self.id_map = {i: gid for gid, i in self.gs_map.items()}
        self.pair_gids = np.asarray(pair_gids, dtype=np.int64).reshape(-1, 2)
        self.pair_labels = np.asarray(pair_labels, dtype=np.int64).reshape(-1)
        self.num_labels = int(len(np.unique(self.pair_labels))) if len(self.pair_labels) else 1
        N = len(molecules)
        rows = self._rows(self.pair_gids)
        self._pair_keys = np.minimum(rows[:, 0], rows[:, 1]) * np.int64(N) + np.maximum(rows[:, 0], rows[:, 1])
        order = np.argsort(self._pair_keys, kind="stable")
        self._sorted_keys, self._sorted_labels = self._pair_keys[order], self.pair_labels[order]
        self.interaction_combo_nxgraph = InteractionGraph(N, rows)
        self.interaction_nxgraphs = None
        if multi_edge_types:
            self.interaction_nxgraphs = {
                int(t): InteractionGraph(N, rows[self.pair_labels == t])
                for t in np.unique(self.pair_labels)}
        self.interaction_num_node_feat = None
        self.graph_feats = None
        self.init = False
        self.train_pairs = self.pair_gids

    def _rows(self, gids):
        gids = np.asarray(gids, dtype=np.int64)
        if gids.size == 0:
            return gids.reshape(-1, 2)
        return self.rows_of_gids(gids)

    def rows_of_gids(self, gids):
        """gs_map[gid] for an int array of any shape through a dense table (built once; graph ids are small
        non-negative integers in every dataset of this path, else the dict is walked); unknown ids raise KeyError
        as the dict lookup of the reference does."""
        gids = np.asarray(gids, dtype=np.int64)
        lut = getattr(self, "_gid_lut", None)
        # (keyed on the map OBJECT and its size: assigning a new gs_map rebuilds the table; code that remaps entries of
        #  the same dict in place calls ``invalidate_gid_table()``)
        if lut is None or getattr(self, "_gid_lut_key", None) != (id(self.gs_map), len(self.gs_map)):
            keys = np.fromiter(self.gs_map.keys(), dtype=np.int64, count=len(self.gs_map))
            if len(keys) and keys.min() >= 0 and keys.max() < 16 * len(keys) + 1024:
                lut = np.full(int(keys.max()) + 1, -1, dtype=np.int64)
                lut[keys] = np.fromiter(self.gs_map.values(), dtype=np.int64, count=len(self.gs_map))
            else:
                lut = False
            self._gid_lut, self._gid_lut_key = lut, (id(self.gs_map), len(self.gs_map))
        if lut is False:
            return np.vectorize(self.gs_map.__getitem__, otypes=[np.int64])(gids)
        if gids.size and (gids.min() < 0 or gids.max() >= lut.shape[0]):
            raise KeyError(int(gids.max() if gids.max() >= lut.shape[0] else gids.min()))
        rows = lut[gids]
        if (rows < 0).any():
            raise KeyError(int(gids[rows < 0].reshape(-1)[0]))
        return rows

    def invalidate_gid_table(self):
        """call after changing entries of ``gs_map`` in place (a re-split / fold change that reassigns rows)"""
        self._gid_lut = None

    def num_graphs(self):
        return len(self.molecules)

    def look_up_graph_by_gid(self, gid):
        return self.molecules.single(int(gid))

    def positive_pair_keys(self):
        return self._sorted_keys

    def labels_of(self, pair_gids):
        """true_label of each known pair (order-insensitive); raises on unknown pairs,
        as ``look_up_pair_by_gids`` does."""
        rows = self._rows(pair_gids)
        N = np.int64(self.num_graphs())
        keys = np.minimum(rows[:, 0], rows[:, 1]) * N + np.maximum(rows[:, 0], rows[:, 1])
        pos = np.searchsorted(self._sorted_keys, keys)
        pos = np.clip(pos, 0, max(len(self._sorted_keys) - 1, 0))
        if len(self._sorted_keys) == 0 or not (self._sorted_keys[pos] == keys).all():
            raise ValueError("pair not found in dataset")
        return self._sorted_labels[pos]

    # utils/data/dataset.py:179-210
    def init_interaction_graph_feats(self, init_method, device, d_init, feat_size=None, seed=0):
        self.interaction_num_node_feat = d_init
        n = self.num_graphs()
        if "rand_init" in init_method:
            g = torch.Generator().manual_seed(seed)
            std = math.sqrt(2.0) * math.sqrt(2.0 / (n + d_init))       # xavier_normal_, gain('relu')
            self.graph_feats = (torch.randn(n, d_init, generator=g) * std).to(device)
        elif "ones_init" in init_method:
            self.graph_feats = torch.ones(n, d_init, device=device)
        elif "one_hot_init" in init_method:
            self.graph_feats = torch.eye(n, device=device)
            self.interaction_num_node_feat = n
        elif init_method == "model_init":
            pass
        elif init_method == "no_init":
            self.graph_feats = torch.zeros(n, d_init, device=device)
        else:
            raise NotImplementedError

    # utils/data/dataset.py:166-177
    def init_interaction_graph_embds(self, device):
        if self.init is True:
            return
        g = self.interaction_combo_nxgraph
        if not isinstance(g.init_x, torch.Tensor):
            g.init_x = self.graph_feats
        g.init_x = g.init_x.to(device)
        self.init = True


class TrainData(object):
    """What ``Model.__init__`` reads from the reference's TorchBiGNNDataset
    (utils/data/dataset.py:27-44)."""

    def __init__(self, dataset, num_node_feat=None, device="cuda:0", num_hyper_edge_feat=0):
        self.dataset = dataset
        self.num_node_feat = dataset.molecules.num_node_feat if num_node_feat is None else num_node_feat
        self.num_hyper_edge_feat = num_hyper_edge_feat
        self.device = device


def one_hot_from_codes(codes, cards=FEATURE_CARDS, device="cpu"):
    """utils/data/representation_node_feat.py:86-97: per attribute one-hot, concatenated
    in attribute order.  ``codes`` int [A, len(cards)]."""
    codes = torch.as_tensor(np.asarray(codes), dtype=torch.int64).to(device)
    offs = torch.as_tensor(np.concatenate([[0], np.cumsum(cards)[:-1]]), dtype=torch.int64).to(device)
    x = torch.zeros((codes.shape[0], int(sum(cards))), dtype=torch.float32, device=device)
    x.scatter_(1, codes + offs.view(1, -1), 1.0)
    return x


def _molecule_sizes(rng, G, shape):
    if shape == "drugbank":
        n = np.rint(rng.lognormal(mean=math.log(24.0), sigma=0.55, size=G))
        return np.clip(n, 2, 457).astype(np.int64)
    if shape == "atoms40":
        return np.clip(np.rint(rng.normal(40.0, 10.0, size=G)), 4, 160).astype(np.int64)
    if shape == "drugcombo":                                  # 3 376 molecules, mean 29.3 atoms, max 155
        n = np.rint(rng.lognormal(mean=math.log(26.5), sigma=0.45, size=G))
        return np.clip(n, 2, 155).astype(np.int64)
    raise ValueError(shape)


def _molecule_edges(rng, sizes):
    """Random recursive spanning tree + ceil(0.065 n) ring-closing edges per molecule,
    returned symmetric, sorted by (molecule, row, col), de-duplicated, LOCAL ids."""
    G = sizes.shape[0]
    node_ptr = np.concatenate([[0], np.cumsum(sizes)])
    A = int(node_ptr[-1])
    mol = np.repeat(np.arange(G, dtype=np.int64), sizes)
    local = np.arange(A, dtype=np.int64) - node_ptr[mol]
    child = local > 0
    parent = np.floor(rng.random(A) * np.maximum(local, 1)).astype(np.int64)
    tm, tu, tv = mol[child], local[child], parent[child]
    k = np.ceil(0.065 * sizes).astype(np.int64)
    em = np.repeat(np.arange(G, dtype=np.int64), k)
    n_e = sizes[em]
    eu = np.floor(rng.random(em.shape[0]) * n_e).astype(np.int64)
    ev = np.floor(rng.random(em.shape[0]) * (n_e - 1)).astype(np.int64)
    ev = ev + (ev >= eu)                                    # ev != eu
    m = np.concatenate([tm, em, tm, em])
    u = np.concatenate([tu, eu, tv, ev])
    v = np.concatenate([tv, ev, tu, eu])
    S = np.int64(sizes.max())
    key = np.unique((m * S + u) * S + v)
    m, u, v = key // (S * S), (key // S) % S, key % S
    edge_ptr = np.concatenate([[0], np.cumsum(np.bincount(m, minlength=G))])
    return np.stack([u, v]), node_ptr, edge_ptr


def _atom_codes(rng, A, cards=FEATURE_CARDS):
    cols = []
    for c in cards:
        p = 1.0 / np.arange(1, c + 1) ** 1.6                 # Zipf-like: C >> N, O > others
        cols.append(rng.choice(c, size=A, p=p / p.sum()))
    return np.stack(cols, axis=1).astype(np.int64)


def _chung_lu_edges(rng, N, E):
    """E distinct undirected pairs with power-law expected degrees (DrugBank-like skew)."""
    w = (np.arange(1, N + 1, dtype=np.float64)) ** -0.55
    rng.shuffle(w)
    p = w / w.sum()
    out = np.zeros((0,), dtype=np.int64)
    while out.shape[0] < E:
        m = int((E - out.shape[0]) * 1.3) + 16
        a = rng.choice(N, size=m, p=p)
        b = rng.choice(N, size=m, p=p)
        keep = a != b
        lo, hi = np.minimum(a, b)[keep], np.maximum(a, b)[keep]
        out = np.unique(np.concatenate([out, lo * np.int64(N) + hi]))
    out = rng.permutation(out)[:E]
    return np.stack([out // N, out % N], axis=1)


def make_synthetic(num_drugs, num_pairs, shape="drugbank", seed=20200614, device="cuda:0",
                   num_edge_types=1, name=None):
    """Deterministic synthetic bi-level dataset.  Graph ids are
    0..num_drugs-1.  ``num_edge_types`` > 1 gives DrugCombo-style labelled pairs."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sizes = _molecule_sizes(rng, num_drugs, shape)
    edge_local, node_ptr, edge_ptr = _molecule_edges(rng, sizes)
    cards = DRUGCOMBO_CARDS if shape == "drugcombo" else FEATURE_CARDS
    codes = _atom_codes(rng, int(node_ptr[-1]), cards)
    x = one_hot_from_codes(codes, cards=cards, device=device)
    store = MoleculeStore(x, edge_local, node_ptr, edge_ptr, range(num_drugs), device, codes=codes, cards=cards)
    max_pairs = num_drugs * (num_drugs - 1) // 2
    pairs = _chung_lu_edges(rng, num_drugs, min(num_pairs, max_pairs))
    if num_edge_types > 1:
        labels = rng.integers(0, num_edge_types, size=pairs.shape[0])
    else:
        labels = np.ones(pairs.shape[0], dtype=np.int64)
    return BiGNNDataset(name or ("synthetic_" + shape), store, pairs, labels,
                        multi_edge_types=num_edge_types > 1)
