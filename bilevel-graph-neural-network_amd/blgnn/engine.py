"""Graph-replayed, two-stream executor of the bi-level forward (the serving runtime around
the kernels).

One forward is ~35 short launches.  ``BiLevelEngine`` captures them once into HIP graphs and
replays them, and it software-pipelines consecutive forwards over two HIP streams: the
molecular sweep of step k+1 (stream L: lower graph [+ the RCCL all-gather]) runs while the
interaction-level pass and the pair scorer of step k (stream U) are still in flight.  The two
halves meet through events and a double-buffered ``init_x`` (step parity selects the buffer),
so no kernel ever reads a buffer another in-flight step is writing.

What runs where mirrors src/train.py:75-108 of the reference: ``lower`` = ``_get_initial_embd``
(every molecule through the lower layers + readout, rows written into the interaction
graph's ``init_x``), ``exchange`` = the all-gather that replaces those row writes across
ranks, ``upper`` = the higher-level layers + LinkPred on one edge-pair batch.
"""

import numpy as np
import torch

from . import ops
from .batch import BatchData
from .layers import NodeEmbedding
from .layers_aggregation import NodeAggregationPairs, _SegmentPool
from .layers_util import act_name, bn_affine, prelu_slopes

__all__ = ["BiLevelEngine", "FusedLowerStack"]

TILE_ROWS = 64


class FusedLowerStack(object):
    """The lower level (GIN x L + avg/sum readout) as ONE launch over tiles of whole molecules
    (``blgnn_gin_stack``), plus the per-layer path for the few molecules larger than a tile.
    Pooled rows are written straight into the interaction graph's ``init_x``.

    Opt-in (``BiLevelEngine(fuse_lower=True)``): with the feature transform on the fp32 matrix
    cores the stack is MFMA-bound either way, and whole-molecule tiles waste ~22 % of the MFMA
    rows on padding and admit only two workgroups per CU, so it measures 228 us against
    5 x 44 us for the per-layer kernels on the DrugBank-shape batch.  It
    removes 4/5 of the activation traffic and will pay once the transform is faster."""

    @staticmethod
    def supported(lower_layers, flags):
        convs, agg = lower_layers[:-1], lower_layers[-1]
        if not convs or len(convs) > 8 or 'gmn' in flags.model:
            return False
        if not isinstance(agg, NodeAggregationPairs) or not isinstance(agg.agg_func, _SegmentPool):
            return False
        n = convs[0].out_dim
        for i, c in enumerate(convs):
            if not isinstance(c, NodeEmbedding) or c.type != 'gin' or c.normalize or c.higher_level:
                return False
            if c.out_dim != n or act_name(c.act) not in ("relu", "identity", "prelu"):
                return False
            if (c.bn and c.bn.training) or c.conv.nn[0].in_features % 4 != 0:
                return False
            if i > 0 and c.conv.nn[0].in_features != n:
                return False
        return n in (64, 128, 256) and convs[0].conv.nn[0].in_features <= 256

    def __init__(self, lower_layers, dataset, gids):
        self.convs, self.agg = lower_layers[:-1], lower_layers[-1]
        self.layers_all = lower_layers
        sizes = dataset.molecules.sizes[[dataset.molecules.gid_to_pos[g] for g in gids]]
        small = [g for g, n in zip(gids, sizes) if n <= TILE_ROWS]
        large = [g for g, n in zip(gids, sizes) if n > TILE_ROWS]
        self.small = self._batch(small, dataset)
        self.large = self._batch(large, dataset)
        self.n = self.convs[0].out_dim
        L = len(self.convs)
        self.pool_cols = [l * self.n for l in range(L)] if self.agg.concat_multi_scale \
            else [-1] * (L - 1) + [0]
        self.width = self.n * (L if self.agg.concat_multi_scale else 1)
        if self.small is not None:
            m = self.small.merge_data['merge']
            ssz = [b - a for a, b in self.small.merge_data['ind_list']]
            tiles = ops.molecule_tiles(ssz, TILE_ROWS)
            self.tile_row = torch.as_tensor(tiles[0]).to(m.x.device)
            self.tile_mol = torch.as_tensor(tiles[1]).to(m.x.device)
            self.rows_small = self.small.interaction_rows_of_batch_graphs()
            m.csr()
            m.segments()

    @staticmethod
    def _batch(gids, dataset):
        if not gids:
            return None
        pairs = np.asarray([[gids[i], gids[min(i + 1, len(gids) - 1)]] for i in range(0, len(gids), 2)])
        return BatchData(pairs, dataset, is_train=False, ignore_pairs=True)

    def _params(self):
        out = []
        for c in self.convs:
            lin1, lin2 = c.conv.nn[0], c.conv.nn[2]
            scale, shift = bn_affine(c._cache, "bn", c.bn)
            out.append({
                "w1": c._cache.get("w1", lambda w: w.t().contiguous(), lin1.weight), "b1": lin1.bias,
                "w2": c._cache.get("w2", lambda w: w.t().contiguous(), lin2.weight), "b2": lin2.bias,
                "act": act_name(c.act), "eps": c._cache.get("eps", lambda e: float(e[0]), c.conv.eps),
                "slope": prelu_slopes(c._cache, "prelu", c.act, c.out_dim), "scale": scale, "shift": shift})
        return out

    def run(self, model, init_x):
        with torch.no_grad():
            if self.small is not None:
                m = self.small.merge_data['merge']
                ops.gin_stack(m.csr(), m.x, self.tile_row, self.tile_mol, m.segments(), self._params(),
                              init_x, self.pool_cols, self.agg.agg_func.kind, row_map=self.rows_small)
            if self.large is not None:                       # molecules bigger than a tile: per-layer kernels
                model.acts = [self.large.merge_data['merge'].x]
                for layer in self.layers_all:
                    model.acts.append(layer(model.acts[-1], self.large, model))


class BiLevelEngine(object):
    def __init__(self, model, lower_batch, upper_batches, graph, sharded=None, world_size=1,
                 use_graph=True, overlap=True, fuse_lower=False, readout_with_upper=None, lower_cus_per_xcd=None):
        from .config import FLAGS
        k = FLAGS.last_lower_lyr_num
        self.model = model
        self.lower_layers = list(model.layers[:k + 1])
        self.upper_layers = list(model.layers[k + 1:-1])      # LoadInteraction ... LinkPred (no Loss)
        self.lower_batch, self.upper_batches = lower_batch, upper_batches
        self.graph, self.sharded, self.world = graph, sharded, world_size
        self.use_graph = use_graph
        self.overlap = overlap and use_graph
        self.nb = len(upper_batches)
        self.parities = 2 if self.overlap else 1
        dev = lower_batch.merge_data['merge'].x.device
        base = graph.init_x
        self.init_x = [base] + [torch.zeros_like(base) for _ in range(self.parities - 1)]
        self.pooled = [None] * self.parities
        self.g_lower = [None] * self.parities
        self.g_upper = [[None] * self.nb for _ in range(self.parities)]
        self.outs = [[None] * self.nb for _ in range(self.parities)]
        gids = lower_batch._graph_gids
        self.fused = None
        if fuse_lower and FusedLowerStack.supported(self.lower_layers, FLAGS) and \
                base.shape[1] == self.lower_layers[0].out_dim * (len(self.lower_layers) - 1
                                                                 if self.lower_layers[-1].concat_multi_scale else 1):
            self.fused = FusedLowerStack(self.lower_layers, lower_batch.dataset, gids)
        rows = lower_batch.interaction_rows_of_batch_graphs()
        self.local_lo = int(rows.min()) if rows.numel() else 0
        self.local_hi = int(rows.max()) + 1 if rows.numel() else 0
        # Two-stream balance knob: the node -> graph readout (17 us, HBM bound) can ride at the head of
        # the interaction-level graph instead of the tail of the molecular sweep; the activations it
        # reads are per-parity buffers, protected by the same event that protects init_x[parity].
        # Single GPU only (with an exchange step the pooled rows must exist before the all-gather).
        # Off by default: at the DrugBank shape it measured 0.223 vs 0.208 ms/step.
        if readout_with_upper is None:
            readout_with_upper = False
        self.readout_with_upper = bool(readout_with_upper and self.overlap and world_size == 1
                                       and self.fused is None and len(self.lower_layers) > 1)
        self.lower_acts = [None] * self.parities
        self.stream_l = torch.cuda.Stream(device=dev) if self.overlap else None
        self.stream_u = torch.cuda.Stream(device=dev) if self.overlap else None
        self.lower_cus_per_xcd = None
        if self.overlap and lower_cus_per_xcd:
            # each half of the step on its own compute units (blgnn/streams.py)
            from .streams import cu_partition_streams
            pair = cu_partition_streams(dev, lower_cus_per_xcd)
            if pair is not None:
                self.stream_l, self.stream_u = pair
                self.lower_cus_per_xcd = int(lower_cus_per_xcd)
        self.ev_lower = [None] * self.parities
        self.ev_upper = [None] * self.parities
        self.step_no = 0

    # ---- the three pieces of one forward -------------------------------------------------
    def lower(self, parity=0, readout=True):
        """-> this rank's pooled rows (a view of init_x: local drugs are a contiguous row range).
        ``readout=False`` stops before the node -> graph readout and keeps the activations for
        ``upper(..., readout=True)``."""
        self.graph.init_x = self.init_x[parity]
        if self.fused is not None:
            self.fused.run(self.model, self.init_x[parity])
            return self.init_x[parity][self.local_lo:self.local_hi]
        with torch.no_grad():
            m = self.model
            m.acts = [self.lower_batch.merge_data['merge'].x]
            for layer in (self.lower_layers if readout else self.lower_layers[:-1]):
                m.acts.append(layer(m.acts[-1], self.lower_batch, m))
            if not readout:
                self.lower_acts[parity] = list(m.acts)
            return self.init_x[parity][self.local_lo:self.local_hi]

    def exchange(self, pooled, parity=0):
        if self.world > 1:                                     # the one collective of the path
            self.sharded.gather_init_x(pooled.clone(), out=self.init_x[parity])

    def upper(self, i, parity=0, readout=False):
        self.graph.init_x = self.init_x[parity]
        with torch.no_grad():
            m, b = self.model, self.upper_batches[i % self.nb]
            if readout:                                        # the sweep's readout, deferred to this graph
                m.acts = list(self.lower_acts[parity])
                self.lower_layers[-1](m.acts[-1], self.lower_batch, m)
            m.acts = [b.merge_data['merge'].x]
            for layer in self.upper_layers:
                m.acts.append(layer(m.acts[-1], b, m))
            return m.acts[-1]

    def eager_step(self, i):
        self.exchange(self.lower(0), 0)
        return self.upper(i, 0)

    # ---- capture ------------------------------------------------------------------------------
    def capture(self):
        """Run every batch once eagerly (host->device id copies and CSR builds must not fall
        into a capture), then record the graphs."""
        for p in range(self.parities):
            for i in range(self.nb):
                self.exchange(self.lower(p), p)
                self.upper(i, p)
        torch.cuda.synchronize()
        if not self.use_graph:
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for p in range(self.parities):
                defer = self.readout_with_upper
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self.pooled[p] = self.lower(p, readout=not defer)
                self.g_lower[p] = g
                for i in range(self.nb):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        self.outs[p][i] = self.upper(i, p, readout=defer)
                    self.g_upper[p][i] = g
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    # ---- steady state ---------------------------------------------------------------------------
    def step(self, i):
        """Enqueue forward number ``i`` (asynchronous); returns the tensor its pair scores land in."""
        if not self.use_graph:
            return self.eager_step(i)
        if not self.overlap:
            self.g_lower[0].replay()
            self.exchange(self.pooled[0], 0)
            self.g_upper[0][i % self.nb].replay()
            return self.outs[0][i % self.nb]
        p = self.step_no & 1
        self.step_no += 1
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(self.stream_l):
            if self.ev_upper[p] is not None:                   # init_x[p] is free once step k-2 scored
                self.stream_l.wait_event(self.ev_upper[p])
            else:
                self.stream_l.wait_stream(cur)
            self.g_lower[p].replay()
            self.exchange(self.pooled[p], p)
            self.ev_lower[p] = torch.cuda.Event()
            self.ev_lower[p].record(self.stream_l)
        with torch.cuda.stream(self.stream_u):
            self.stream_u.wait_event(self.ev_lower[p])
            self.g_upper[p][i % self.nb].replay()
            self.ev_upper[p] = torch.cuda.Event()
            self.ev_upper[p].record(self.stream_u)
        return self.outs[p][i % self.nb]

    def synchronize(self):
        if self.overlap:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream_l)
            cur.wait_stream(self.stream_u)
        torch.cuda.synchronize()
