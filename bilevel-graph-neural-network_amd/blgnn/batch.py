"""``BatchData`` -- the edge-pair mini-batch handed to every layer; drop-in for the
reference's src/batch.py on this path.

Same constructor arguments and the same attributes the layers read
(``merge_data``, ``merge_higher_level``, ``batch_gids``, ``pair_list``,
``interaction_combo_nxgraph``, ``dataset``, ``is_train``, ``batch_interaction_inds``).
The per-pair python work of src/batch.py:105-144 (look up graph, nx -> tensors,
dedup, merge) is one vectorised gather from the pre-tensorised ``MoleculeStore``;
negative sampling (src/batch.py:61-103) is vectorised rejection sampling with the
same acceptance rules.
"""

import numpy as np
import torch

from .config import FLAGS

__all__ = ["BatchData", "SharedMolecules", "ordered_unique"]


def ordered_unique(a):
    """Unique values in order of first appearance (the insertion order of the
    reference's ``graphs_in_batch`` dict: src/batch.py:131-136)."""
    flat = np.asarray(a).reshape(-1)
    _, idx = np.unique(flat, return_index=True)
    return flat[np.sort(idx)]


class _Pair(object):
    """Lazy view of one pair's label / prediction (utils/data/graph.py GraphPair)."""

    __slots__ = ("_batch", "_i")

    def __init__(self, batch, i):
        self._batch, self._i = batch, i

    @property
    def true_label(self):
        return int(self._batch._labels[self._i])

    def get_link_pred(self):
        preds = getattr(self._batch, "pair_preds", None)
        if preds is None:
            raise ValueError('Must call assign_link_pred before calling this')
        if self._batch._preds_host is None:
            self._batch._preds_host = preds.detach().cpu().numpy()     # ONE sync for the batch
        p = self._batch._preds_host[self._i]
        return float(p[0]) if p.shape[0] == 1 else p


class _PairList(object):
    """``pair_list`` of the reference (src/batch.py:112-114: one GraphPair per row of the batch) as a sequence of lazy
    views: a 4 096-pair training batch does not build 4 096 python objects nobody reads."""

    __slots__ = ("_batch", "_n")

    def __init__(self, batch, n):
        self._batch, self._n = batch, int(n)

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [_Pair(self._batch, k) for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return _Pair(self._batch, i)

    def __iter__(self):
        return (_Pair(self._batch, k) for k in range(self._n))

    def __eq__(self, other):
        return list(self) == other if isinstance(other, list) else NotImplemented


class SharedMolecules(object):
    """``graph``: GraphData of the distinct molecules of a pair batch; ``row_map`` int32 [n] (device): merged atom ->
    its row in ``graph``."""

    __slots__ = ("graph", "row_map")

    def __init__(self, graph, row_map):
        self.graph, self.row_map = graph, row_map


class BatchData(object):
    def __init__(self, batch_gids, dataset, sampled_gids=None, curr_sample_edge_type=None,
                 is_train=True, ignore_pairs=False, enforce_negative_sampling=True,
                 unique_graphs=True, subgraph=None, rng=None):
        self.dataset = dataset
        self.is_train = is_train
        if isinstance(batch_gids, torch.Tensor):
            batch_gids = batch_gids.cpu().numpy()
        self.batch_gids = np.asarray(batch_gids).astype('int').reshape(-1, 2)
        if hasattr(dataset, "interaction_combo_nxgraph"):
            self.interaction_combo_nxgraph = dataset.interaction_combo_nxgraph
            if not is_train:
                self._snapshot_init_x()
        self.unique_graphs = unique_graphs
        self.positive_pair_gids = self.batch_gids
        self.ignore_pairs = ignore_pairs
        self.curr_sample_edge_type = curr_sample_edge_type
        self._labels = None
        self._preds_host = None
        if getattr(FLAGS, "negative_sample", False) and is_train and not ignore_pairs:
            assert sampled_gids is not None
            self.sampled_gids = np.asarray(sampled_gids)
            self.sample_negative_pairs(enforce_negative_sampling,
                                       rng if rng is not None else np.random.default_rng())
        self.merge_data, self.pair_list = self._merge_into_one_graph(self.batch_gids)
        self.batch_interaction_inds = self._rows_of(self.batch_gids.reshape(-1)).tolist()
        self.merge_higher_level = {}
        if subgraph is not None:
            raise NotImplementedError("sampled subgraphs (use_sampled_subgraph) are outside the hot path")

    def _rows_of(self, gids):
        """gs_map[gid] of an int array (dense table of the dataset when it has one)."""
        if hasattr(self.dataset, "rows_of_gids"):
            return self.dataset.rows_of_gids(gids)
        return np.vectorize(self.dataset.gs_map.__getitem__, otypes=[np.int64])(np.asarray(gids, dtype=np.int64))

    # -- negatives: src/batch.py:61-103 ------------------------------------------------
    def sample_negative_pairs(self, enforce_negative, rng):
        gs_map = self.dataset.gs_map
        want = len(self.batch_gids) * int(getattr(FLAGS, "num_negative_samples", 1))
        pool = self.sampled_gids if getattr(FLAGS, "enforce_sampling_amongst_same_graphs", True) \
            else np.asarray(list(gs_map.keys()))
        n = len(self.sampled_gids)
        if getattr(FLAGS, "enforce_sampling_amongst_same_graphs", True):
            want = int(min(want, n * (n - 1) // 2 - len(self.batch_gids)))
        pos = self.dataset.positive_pair_keys()
        N = np.int64(self.dataset.num_graphs())
        rows = self._rows_of
        bp = rows(self.positive_pair_gids)
        pos = np.union1d(pos, np.minimum(bp[:, 0], bp[:, 1]) * N + np.maximum(bp[:, 0], bp[:, 1]))
        # rounds of candidates, accepted in candidate order: not a self pair, not a known positive, not drawn before
        # (the acceptance rules of src/batch.py:75-98, on whole arrays instead of one candidate per python iteration)
        chosen = np.zeros((0, 2), dtype=np.int64)
        seen = np.zeros(0, dtype=np.int64)
        cursor = 0
        while len(chosen) < want:
            m = max(2 * (want - len(chosen)), 16)
            a = self.sampled_gids[(cursor + np.arange(m)) % n]
            cursor += m
            b = rng.choice(pool, size=m)
            if enforce_negative:
                ra, rb = rows(a), rows(b)
                key = np.minimum(ra, rb) * N + np.maximum(ra, rb)
                ok = (a != b) & ~np.isin(key, pos) & ~np.isin(key, seen)
                first = np.zeros(m, dtype=bool)
                first[np.unique(key, return_index=True)[1]] = True          # first occurrence of a key in this round
                ok &= first
                # (a key's first occurrence may itself be rejected only for being a self pair / positive / seen, and then
                #  every later occurrence of that key is rejected for the same reason)
            else:
                key = np.arange(m)
                ok = np.ones(m, dtype=bool)
            take = np.flatnonzero(ok)[:want - len(chosen)]
            chosen = np.concatenate((chosen, np.stack((a[take], b[take]), axis=1).astype(np.int64)))
            if enforce_negative:
                seen = np.concatenate((seen, key[take]))
        if len(chosen):
            self.negative_pair_gids = chosen
            self._neg_start = len(self.batch_gids)
            self.batch_gids = np.concatenate((self.batch_gids, self.negative_pair_gids))

    # -- merge: src/batch.py:105-144 + src/merged_graph.py --------------------------------
    def _merge_into_one_graph(self, batch_gids):
        store = self.dataset.molecules
        if self.unique_graphs:
            gids = ordered_unique(batch_gids).tolist()
        else:
            gids = batch_gids.reshape(-1).tolist()
        merged = store.merge(gids, self.unique_graphs)
        self._graph_gids = gids
        if self.ignore_pairs:
            return merged, []
        labels = self.dataset.labels_of(self.positive_pair_gids)
        n_neg = len(batch_gids) - len(self.positive_pair_gids)
        self._labels = np.concatenate([labels, np.zeros(n_neg, dtype=np.int64)])
        return merged, _PairList(self, len(batch_gids))

    def shared_molecules(self):
        """Pair batches (``unique_graphs=False``: src/config.py:258 for the GMN model) hold one copy of a molecule per
        pair it occurs in.  -> ``SharedMolecules`` (the distinct molecules merged once + the row of every merged atom
        in that graph), or None for a unique-graph batch.  Layers whose input does not yet depend on the pair partner
        (everything in front of the first cross-graph attention) compute on the distinct molecules and replicate rows."""
        sm = getattr(self, "_shared", None)
        if sm is None:
            sm = False
            if not self.unique_graphs and self._graph_gids:
                gids = np.asarray(self._graph_gids)
                um = self.dataset.molecules.merge(ordered_unique(gids).tolist(), True)
                inst_u = np.vectorize(um["gids_to_batch_ind"].get)(gids).astype(np.int64)
                sizes = np.asarray(um["graph_sizes"], dtype=np.int64)
                u_off = np.concatenate([[0], np.cumsum(sizes)])
                inst_sizes = sizes[inst_u]
                inst_off = np.concatenate([[0], np.cumsum(inst_sizes)])
                n = int(inst_off[-1])
                assert n == self.merge_data['merge'].x.shape[0]
                # row of merged atom r (instance i) in the distinct graph = u_off[inst_u[i]] + (r - inst_off[i]): expanded
                # on the device from two [G] arrays
                dev = self.merge_data['merge'].x.device
                host = torch.as_tensor(np.stack([u_off[inst_u] - inst_off[:-1], inst_sizes]))
                delta_t, isz_t = host.to(dev, non_blocking=True).unbind(0)
                row_map = torch.repeat_interleave(delta_t, isz_t, output_size=n) + torch.arange(n, dtype=torch.int64, device=dev)
                sm = SharedMolecules(um["merge"], row_map.to(torch.int32))
            self._shared = sm
        return sm or None

    def prepare(self):
        """Build every device-side layout a forward derives from the host-side batch -- CSR, segment offsets, the int32
        batch vector, the distinct-molecule graph + row map and (pair batches) the cross-graph attention's size-class pair
        order -- NOW, on the current stream.  Each is otherwise built on first use inside the forward, where its
        host-to-device copy makes the host wait for the kernels already enqueued: a loop that builds batch k + 1 on a side
        stream while batch k runs calls this at build time (and, as with any tensor allocated on one stream and read on
        another, keeps the batch alive until the stream that READS it is done with it -- or ``record_stream``s its
        tensors: bench.py's ``pipelined_step``).  -> self"""
        m = self.merge_data['merge']
        m.csr()
        m.segments()
        m.row_graph()
        sm = self.shared_molecules()
        if sm is not None:
            sm.graph.csr()
            sm.graph.segments()
        if not self.unique_graphs and len(self.merge_data['ind_list']) % 2 == 0:
            from .layers_gmn import GMNPropagator
            GMNPropagator._max_nodes(self)
            GMNPropagator._pair_order(self, m.x.device)
        return self

    def interaction_rows_of_batch_graphs(self):
        """gs_map[gid] for every graph of the merged batch, int32 on the device."""
        rows = getattr(self, "_graph_rows", None)
        if rows is None:
            host = self._rows_of(np.asarray(self._graph_gids, dtype=np.int64)).astype(np.int32)
            rows = torch.as_tensor(host).to(self.merge_data['merge'].x.device)
            self._graph_rows = rows
            # contiguous ascending run of rows (a sweep over the drugs in gs_map order): the readout can
            # write its result straight into that slice of init_x
            self._graph_row_range = (int(host[0]), int(host[-1]) + 1) if (
                len(host) > 0 and np.array_equal(host, np.arange(host[0], host[0] + len(host)))) else None
        return rows

    def interaction_row_range(self):
        """(lo, hi) when the batch graphs are rows lo..hi-1 of the interaction graph in order, else None."""
        self.interaction_rows_of_batch_graphs()
        return self._graph_row_range

    def true_labels(self):
        return torch.as_tensor(self._labels).to(self.merge_data['merge'].x.device)

    # -- eval-time snapshot of init_x: src/batch.py:146-162 ---------------------------------
    def _snapshot_init_x(self):
        g = self.interaction_combo_nxgraph
        if isinstance(getattr(g, "init_x", None), torch.Tensor):
            self.save_init_x = g.init_x
            g.init_x = self.save_init_x.clone()

    def restore_interaction_nxgraph(self):
        g = self.dataset.interaction_combo_nxgraph
        if isinstance(g.init_x, torch.Tensor):
            g.init_x = self.save_init_x.detach() if hasattr(self, "save_init_x") else g.init_x.detach()
