"""Graph containers and layout builders for the hot path.

Mirrors, for this path only, what the reference gets from
``torch_geometric.data.Data`` and ``src/merged_graph.py`` -- plus the CSR /
segment-offset layout the HIP kernels consume, built once per batch (molecules)
or once per fold (interaction graph) instead of once per layer call.
"""

import numpy as np
import torch

from . import ops

__all__ = ["Ranges", "GraphData", "MoleculeStore", "InteractionGraph", "merge_from_data_list",
           "to_undirected_host"]


def to_undirected_host(edges, num_nodes):
    """``to_undirected`` of model/layers_util.py:166 on the host: both directions,
    sorted by (row, col), duplicates removed.  ``edges`` is an int array [E, 2]."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    if e.shape[0] == 0:
        return np.zeros((2, 0), dtype=np.int64)
    row = np.concatenate([e[:, 0], e[:, 1]])
    col = np.concatenate([e[:, 1], e[:, 0]])
    key = np.unique(row * np.int64(num_nodes) + col)
    return np.stack([key // num_nodes, key % num_nodes], axis=0)


class Ranges(object):
    """``ind_list`` / ``edge_ind_list`` of src/merged_graph.py:63-64 -- the list of (start, end) ranges of the batch's graphs
    -- kept as its offsets array: indexing, iteration, len() and == behave like the reference's list of tuples, and callers
    that want all sizes at once (``sizes()``) do not walk 16 384 tuples per batch."""

    __slots__ = ("offsets",)

    def __init__(self, offsets):
        self.offsets = np.asarray(offsets, dtype=np.int64)

    def __len__(self):
        return max(self.offsets.shape[0] - 1, 0)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        return (int(self.offsets[i]), int(self.offsets[i + 1]))

    def __iter__(self):
        o = self.offsets.tolist()
        return iter(zip(o[:-1], o[1:]))

    def __eq__(self, other):
        return list(self) == list(other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __repr__(self):
        return "Ranges(%r)" % (list(self),)

    def sizes(self):
        return np.diff(self.offsets)


class GraphData(object):
    """The fields of ``torch_geometric.data.Data`` this path touches
    (model/layers.py:44-50, model/layers_aggregation.py:31, src/merged_graph.py:92)
    plus cached device layouts."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, batch=None, num_nodes=None,
                 num_graphs=None):
        self.x, self.edge_index, self.edge_attr, self.batch = x, edge_index, edge_attr, batch
        self._num_nodes = num_nodes
        self.num_graphs = num_graphs
        # True when the builder of this object guarantees a symmetric edge set (to_undirected / a checked MoleculeStore):
        # the CSR then skips its device-side symmetry check (ops.CSR.require_symmetric)
        self.symmetric_by_construction = False
        self._csr = None
        self._seg = None
        self._batch32 = None
        # compact form of x when it is a concatenation of one-hot blocks: int8 [n, 8] value indices and the
        # first column of every block (+ total width); the first lower layer can read these instead of x
        self.codes = None
        self.code_offsets = None

    @property
    def num_nodes(self):
        if self._num_nodes is not None:
            return self._num_nodes
        return self.x.shape[0]

    def is_undirected(self):
        """model/layers_util.py:133 -- symmetric edge set check (host side, tests only)."""
        ei = self.edge_index.cpu().numpy()
        n = self.num_nodes
        fwd = np.unique(ei[0] * np.int64(n) + ei[1])
        bwd = np.unique(ei[1] * np.int64(n) + ei[0])
        return fwd.shape == bwd.shape and bool((fwd == bwd).all())

    def csr(self, symmetric=True):
        if self._csr is None:
            self._csr = ops.csr_from_edge_index(self.edge_index, self.num_nodes, symmetric)
            if symmetric and self.symmetric_by_construction:
                self._csr._symmetric = True
        return self._csr

    def segments(self):
        """Graph segment offsets seg[G+1] (int32, device) of the sorted batch vector."""
        if self._seg is None:
            self._seg = ops.segments_from_batch(self.batch, self.num_graphs)
        return self._seg

    def row_graph(self):
        """The sorted batch vector as int32 (what the fused-readout epilogue indexes by)."""
        if self._batch32 is None:
            self._batch32 = self.batch.to(torch.int32).contiguous()
        return self._batch32

    def contiguous(self):
        return self


def merge_from_data_list(graphs_in_batch, unique_graphs=True):
    """src/merged_graph.py:27-96 (``MergedGraphData.from_data_list``): block-diagonal
    merge of per-molecule GraphData.  Same return dict, same key names."""
    assert (unique_graphs and isinstance(graphs_in_batch, dict)) or \
        (not unique_graphs and isinstance(graphs_in_batch, list))
    data_list = list(graphs_in_batch.values()) if unique_graphs else graphs_in_batch
    sizes = [d.num_nodes for d in data_list]
    esizes = [d.edge_index.shape[1] for d in data_list]
    node_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    edge_off = np.concatenate([[0], np.cumsum(esizes)]).astype(np.int64)
    dev = data_list[0].x.device
    x = torch.cat([d.x for d in data_list], dim=0)
    ei = torch.cat([d.edge_index + int(node_off[i]) for i, d in enumerate(data_list)], dim=1)
    batch = torch.repeat_interleave(
        torch.arange(len(data_list), device=dev), torch.as_tensor(sizes, device=dev))
    merged = GraphData(x=x, edge_index=ei, batch=batch, num_graphs=len(data_list))
    merged.symmetric_by_construction = all(getattr(d, "symmetric_by_construction", False) for d in data_list)
    merged._seg = torch.as_tensor(node_off, dtype=torch.int32, device=dev)
    ind_list = [(int(node_off[i]), int(node_off[i + 1])) for i in range(len(data_list))]
    edge_ind_list = [(int(edge_off[i]), int(edge_off[i + 1])) for i in range(len(data_list))]
    if unique_graphs:
        gids_to_batch_ind = {gid: i for i, gid in enumerate(graphs_in_batch.keys())}
        graph_sizes = np.asarray(sizes, dtype=np.int64)
    else:
        gids_to_batch_ind = graph_sizes = None
    return {"merge": merged, "ind_list": ind_list, "edge_ind_list": edge_ind_list,
            "gids_to_batch_ind": gids_to_batch_ind, "graph_sizes": graph_sizes}


class MoleculeStore(object):
    """Every molecule of the dataset tensorised ONCE (features + symmetric sorted
    local edge lists), resident on the device.  Replaces the per-batch
    nx -> tensor conversion of src/batch.py:131-139 / model/layers_util.py:100-135;
    a batch is then a vectorised gather of slices."""

    def __init__(self, x_all, edge_local, node_ptr, edge_ptr, gids, device, codes=None, cards=None):
        """``x_all`` [A, F] node features of all molecules back to back (tensor or
        ndarray), ``edge_local`` int64 [2, E] per-molecule LOCAL node ids, symmetric and
        sorted by (row, col) within each molecule, ``node_ptr`` / ``edge_ptr`` the
        molecule offsets into them, ``gids`` the graph ids in the same order."""
        self.gids = [int(g) for g in gids]
        self.gid_to_pos = {g: i for i, g in enumerate(self.gids)}
        self.node_ptr = np.asarray(node_ptr, dtype=np.int64)
        self.edge_ptr = np.asarray(edge_ptr, dtype=np.int64)
        self.sizes = np.diff(self.node_ptr)
        self.esizes = np.diff(self.edge_ptr)
        self.device = torch.device(device)
        self.x = torch.as_tensor(x_all, dtype=torch.float32).to(self.device)
        el = np.asarray(edge_local.cpu() if isinstance(edge_local, torch.Tensor) else edge_local, dtype=np.int64).reshape(2, -1)
        # the contract above, checked ONCE per store on the host: every batch merged from it is symmetric by construction
        # (the backward kernels reuse the forward CSR as its transpose; ops.CSR.require_symmetric)
        mol = np.repeat(np.arange(len(self.esizes), dtype=np.int64), self.esizes)
        width = int(self.sizes.max()) + 1 if len(self.sizes) else 1
        fwd = np.sort((mol * width + el[0]) * width + el[1])
        bwd = np.sort((mol * width + el[1]) * width + el[0])
        self.symmetric = bool(fwd.shape == bwd.shape and (fwd == bwd).all())
        self.edge_local = torch.as_tensor(el).to(self.device)
        self.num_node_feat = int(self.x.shape[1])
        self._ptr_dev = None
        self.codes, self.code_offsets = None, None
        if codes is not None and cards is not None and len(cards) <= 8:
            c = np.zeros((np.asarray(codes).shape[0], 8), dtype=np.int8)
            c[:, :len(cards)] = np.asarray(codes, dtype=np.int8)
            self.codes = torch.as_tensor(c).to(self.device)
            self.code_offsets = [0] + np.cumsum([int(v) for v in cards]).tolist()

    @classmethod
    def from_lists(cls, xs, edge_indices, gids, device):
        sizes = np.asarray([x.shape[0] for x in xs], dtype=np.int64)
        esizes = np.asarray([e.shape[1] for e in edge_indices], dtype=np.int64)
        x_all = np.concatenate([np.asarray(x, dtype=np.float32) for x in xs], axis=0)
        e_all = np.concatenate([np.asarray(e, dtype=np.int64).reshape(2, -1) for e in edge_indices], axis=1)
        return cls(x_all, e_all, np.concatenate([[0], np.cumsum(sizes)]),
                   np.concatenate([[0], np.cumsum(esizes)]), gids, device)

    def __len__(self):
        return len(self.gids)

    def single(self, gid):
        p = self.gid_to_pos[gid]
        a, b = int(self.node_ptr[p]), int(self.node_ptr[p + 1])
        ea, eb = int(self.edge_ptr[p]), int(self.edge_ptr[p + 1])
        g = GraphData(x=self.x[a:b], edge_index=self.edge_local[:, ea:eb])
        g.symmetric_by_construction = self.symmetric
        return g

    def merge(self, gid_list, unique_graphs=True):
        """Vectorised equivalent of ``merge_from_data_list`` for molecules of this
        store, in the order given (duplicates allowed when not ``unique_graphs``)."""
        pos = np.asarray([self.gid_to_pos[g] for g in gid_list], dtype=np.int64)
        G = pos.shape[0]
        sizes, esizes = self.sizes[pos], self.esizes[pos]
        node_off = np.concatenate([[0], np.cumsum(sizes)])
        edge_off = np.concatenate([[0], np.cumsum(esizes)])
        Nn, nnz = int(node_off[-1]), int(edge_off[-1])
        dev = self.device
        # The gather indices are built ON THE DEVICE from five [G]-sized arrays (the host walked ~1.5 M elements per
        # 8 192-pair batch in numpy: 20 of the 25 ms a batch took to build).  For merged node r of graph g:
        # src = node_ptr[pos[g]] + (r - node_off[g]); edges likewise, shifted by node_off[g].
        if self._ptr_dev is None:
            self._ptr_dev = (torch.as_tensor(self.node_ptr).to(dev), torch.as_tensor(self.edge_ptr).to(dev))
        node_ptr_t, edge_ptr_t = self._ptr_dev
        host = torch.as_tensor(np.stack([pos, sizes, esizes, node_off[:-1], edge_off[:-1]]))      # [5, G] in one copy
        pos_t, sizes_t, esizes_t, node_off_t, edge_off_t = host.to(dev, non_blocking=True).unbind(0)
        graphs = torch.arange(G, dtype=torch.int64, device=dev)
        g_of_node = torch.repeat_interleave(graphs, sizes_t, output_size=Nn)
        src_node_t = (node_ptr_t[pos_t] - node_off_t)[g_of_node] + torch.arange(Nn, dtype=torch.int64, device=dev)
        g_of_edge = torch.repeat_interleave(graphs, esizes_t, output_size=nnz)
        src_edge_t = (edge_ptr_t[pos_t] - edge_off_t)[g_of_edge] + torch.arange(nnz, dtype=torch.int64, device=dev)
        x = self.x.index_select(0, src_node_t)
        ei = self.edge_local.index_select(1, src_edge_t) + node_off_t[g_of_edge]
        merged = GraphData(x=x, edge_index=ei, batch=g_of_node, num_graphs=G)
        merged.symmetric_by_construction = self.symmetric
        merged._seg = torch.as_tensor(node_off, dtype=torch.int32).to(dev, non_blocking=True)
        if self.codes is not None:
            merged.codes = self.codes.index_select(0, src_node_t)
            merged.code_offsets = self.code_offsets
        ind_list = Ranges(node_off)
        edge_ind_list = Ranges(edge_off)
        if unique_graphs:
            gids_to_batch_ind = {g: i for i, g in enumerate(gid_list)}
            graph_sizes = sizes.copy()
        else:
            gids_to_batch_ind = graph_sizes = None
        return {"merge": merged, "ind_list": ind_list, "edge_ind_list": edge_ind_list,
                "gids_to_batch_ind": gids_to_batch_ind, "graph_sizes": graph_sizes}


class InteractionGraph(object):
    """The interaction-level graph (``dataset.interaction_combo_nxgraph`` of the
    reference, utils/data/dataset.py:72-103) as arrays: ``edges`` int [E,2]
    undirected pairs over node ids 0..N-1, ``init_x`` [N, D] device tensor or
    None.  The symmetric sorted ``edge_index`` and its CSR are built once and
    reused by every forward (the reference rebuilds them per forward:
    model/layers_load_interaction_graph.py:11-20)."""

    def __init__(self, num_nodes, edges, init_x=None):
        self.num_nodes = int(num_nodes)
        self.edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
        self.init_x = init_x
        self._data = None

    def number_of_nodes(self):
        return self.num_nodes

    def number_of_edges(self):
        return int(self.edges.shape[0])

    def graph_data(self, device):
        if self._data is None or self._data.edge_index.device != torch.device(device):
            ei = to_undirected_host(self.edges, self.num_nodes)
            self._data = GraphData(edge_index=torch.as_tensor(ei).to(device),
                                   num_nodes=self.num_nodes)
            self._data.symmetric_by_construction = True          # to_undirected_host
        self._data.x = self.init_x
        return self._data
