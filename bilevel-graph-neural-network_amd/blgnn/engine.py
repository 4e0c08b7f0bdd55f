"""Graph-replayed, two-stream executor of the bi-level forward (the serving runtime around
the kernels).

One forward is ~35 short launches.  ``BiLevelEngine`` captures them once into HIP graphs and
replays them, and it software-pipelines consecutive forwards over two HIP streams: the
molecular sweep of step k+1 (stream L: lower graph [+ the RCCL all-gather]) runs while the
interaction-level pass and the pair scorer of step k (stream U) are still in flight.  The two
halves meet through events and a double-buffered ``init_x`` (step parity selects the buffer),
so no kernel ever reads a buffer another in-flight step is writing.

The pair batch is DATA, not structure: the upper graph reads the rows it scores from a static
``[2, B]`` int64 device buffer per parity, which ``step(pairs)`` fills with one asynchronous
host-to-device copy, so ONE captured graph serves every batch the sampler produces
(src/sampler.py:120-131 hands the reference a fresh ``[B, 2]`` gid batch per iteration).

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

    Opt-in (``BiLevelEngine(fuse_lower=True)``, ``bench.py --fuse-lower``).  With the f16x3 transform
    (d = 128, F_in <= 128: ``blgnn_gin_stack_h3``) the launch takes 110 us for the DrugBank-shape sweep
    against 5 x 29 us + 18 us of per-layer kernels + readout; other shapes take the exact-f32 stack (MFMA
    bound: 228 us against 5 x 44 us)."""

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
        from .config import FLAGS
        h3 = (getattr(FLAGS, "feature_transform", "f16x3") == "f16x3" and self.n == 128
              and self.convs[0].conv.nn[0].in_features <= 128)
        out = []
        for c in self.convs:
            lin1, lin2 = c.conv.nn[0], c.conv.nn[2]
            scale, shift = bn_affine(c._cache, "bn", c.bn)
            lay = {"b1": lin1.bias, "b2": lin2.bias,
                   "act": act_name(c.act), "eps": c._cache.get("eps", lambda e: float(e[0]), c.conv.eps),
                   "slope": prelu_slopes(c._cache, "prelu", c.act, c.out_dim), "scale": scale, "shift": shift}
            if h3:      # f16x3 stack (blgnn_gin_stack_h3): range-normalised two-piece fp16 weight images
                lay["w1p"] = c._cache.get("w1h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin1.weight)
                lay["w2p"] = c._cache.get("w2h", lambda w: ops.pack_weight_h3(w.t().contiguous()), lin2.weight)
                lay["N"] = self.n
            else:
                lay["w1"] = c._cache.get("w1", lambda w: w.t().contiguous(), lin1.weight)
                lay["w2"] = c._cache.get("w2", lambda w: w.t().contiguous(), lin2.weight)
            out.append(lay)
        return out

    def run(self, model, init_x):
        with torch.no_grad():
            if self.small is not None:
                m = self.small.merge_data['merge']
                ops.gin_stack(m.csr(), m.x, self.tile_row, self.tile_mol, m.segments(), self._params(),
                              init_x, self.pool_cols, self.agg.agg_func.kind, row_map=self.rows_small)
            if self.large is not None:                       # molecules bigger than a tile: per-layer kernels
                # (on the same stream: forked onto a side stream inside the captured graph the chain lost the
                # launch stream's CU mask and the step got slower, 0.25 vs 0.21 ms)
                model.acts = [self.large.merge_data['merge'].x]
                for layer in self.layers_all:
                    model.acts.append(layer(model.acts[-1], self.large, model))


class BiLevelEngine(object):
    """``upper_batches``: one ``BatchData`` (the template whose layout every scored batch shares: B
    pairs, evaluation mode) or a list of them; ``step(i)`` with an int scores list entry ``i % len``,
    ``step(pairs)`` with an int array ``[B, 2]`` of graph ids scores those pairs."""

    def __init__(self, model, lower_batch, upper_batches, graph, sharded=None, world_size=1,
                 use_graph=True, overlap=True, fuse_lower=None, readout_with_upper=None, lower_cus_per_xcd=None,
                 exchange_transformed=None, prune_last=None, parity_streams=False, serialize=None, shard_upper=None, collective=None):
        from .config import FLAGS
        k = FLAGS.last_lower_lyr_num
        self.model = model
        # the last interaction-level GAT layer restricted to the rows the scored pairs read: default when sharded
        # (a rank scores B / N pairs out of all the interaction rows), opt-in on one GPU
        self.prune_last = bool(world_size > 1 if prune_last is None else prune_last)
        self.lower_layers = list(model.layers[:k + 1])
        self.upper_layers = list(model.layers[k + 1:-1])      # LoadInteraction ... LinkPred (no Loss)
        if isinstance(upper_batches, BatchData):
            upper_batches = [upper_batches]
        self.lower_batch, self.upper_batches = lower_batch, list(upper_batches)
        self.graph, self.sharded, self.world = graph, sharded, world_size
        self.use_graph = use_graph
        self.overlap = overlap and use_graph
        self.nb = len(self.upper_batches)
        self.parities = 2 if self.overlap else 1
        dev = lower_batch.merge_data['merge'].x.device
        self.device = dev
        base = graph.init_x
        self.init_x = [base] + [torch.zeros_like(base) for _ in range(self.parities - 1)]
        self.pooled = [None] * self.parities
        self.g_lower = [None] * self.parities
        self.g_upper = [None] * self.parities
        self.outs = [None] * self.parities
        # ---- the pair batch as data: row ids of the pairs in pinned host memory and static device buffers
        self.template = self.upper_batches[0]
        pred = self.upper_layers[-1]
        host = [np.ascontiguousarray(pred.pair_row_ids(b).T) for b in self.upper_batches]      # [2, B] each
        self.B = int(host[0].shape[1])
        if any(h.shape != (2, self.B) for h in host):
            raise ValueError("blgnn: every upper batch of an engine must score the same number of pairs")
        self.host_ids = [torch.as_tensor(h).pin_memory() for h in host]
        self.stage_ids = [torch.empty((2, self.B), dtype=torch.int64).pin_memory() for _ in range(self.parities)]
        self.stage_np = [t.numpy() for t in self.stage_ids]                  # the same pinned bytes, as numpy views
        self.ids_dev = [torch.zeros((2, self.B), dtype=torch.int64, device=dev) for _ in range(self.parities)]
        ds = self.template.dataset
        table = ds.gs_map if FLAGS.higher_level_layers else None
        self._row_lut = None
        if table is not None and pred.batch_unique_graphs:
            keys = np.fromiter(table.keys(), dtype=np.int64, count=len(table))
            if keys.size and keys.min() >= 0 and keys.max() < 8 * keys.size + 1024:
                lut = np.full(int(keys.max()) + 1, -1, dtype=np.int64)
                lut[keys] = np.fromiter(table.values(), dtype=np.int64, count=len(table))
                self._row_lut = lut
        gids = lower_batch._graph_gids
        self.fused = None
        if fuse_lower is None:
            # Opt-in.  The f16x3 stack (d = 128) needs 110 us for the DrugBank-shape sweep against 5 x 29 + 18 us
            # for the per-layer kernels + readout, but the per-layer chain of the few molecules larger than a tile
            # then runs behind it and the PIPELINED step is slower (0.206 vs 0.190 ms: DESIGN.md section 4).
            fuse_lower = False
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
        # The interaction-level pass is short (2.1 ms alone at configs[3]) and the next step's sweep long (20.7 ms); beside
        # the sweep its kernels wait for workgroup slots the sweep's 62 500 workgroups keep refilling (kernel trace:
        # k_link_head 0.19 ms alone, 5-9 ms on average beside the sweep).  The short pass runs on a high-priority stream; the
        # measured effect is small (22.3 vs 22.0 ms/step on a cool device, 31 vs 40 on one that was throttling), it costs nothing.
        import os
        prio = -1 if os.environ.get("BLGNN_ENGINE_UPPER_PRIORITY", "1") != "0" else 0
        self.stream_l = torch.cuda.Stream(device=dev) if self.overlap else None
        self.stream_u = torch.cuda.Stream(device=dev, priority=prio) if self.overlap else None
        self.lower_cus_per_xcd = None
        self._masked = None
        if self.overlap and lower_cus_per_xcd:
            # each half of the step on its own compute units (blgnn/streams.py); the engine owns the pair
            from .streams import cu_partition_streams
            pair = cu_partition_streams(dev, lower_cus_per_xcd)
            if pair is not None:
                self._masked = pair
                self.stream_l, self.stream_u = pair.first, pair.second
                self.lower_cus_per_xcd = pair.cus
        # What crosses the wire.  Default with N > 1 and a GAT first interaction layer: the rank applies that
        # layer's ROW-WISE half (h = pooled @ W, both attention scores) to its own pooled rows and the all-gather
        # carries [N_r, d + 2] instead of [N_r, D_pool] (5x fewer bytes with a five-scale readout), and the
        # D_pool-wide GEMM is no longer replicated on every rank.  Same kernels on the same rows: the scores are
        # bit for bit those of the replicated evaluation (``exchange_transformed=True`` at one rank runs this
        # path without a collective: tests/test_gpu_model.py).
        first = self.upper_layers[1] if len(self.upper_layers) > 2 else None
        can = (isinstance(first, NodeEmbedding) and first.type == 'gat' and first.higher_level
               and not self.readout_with_upper)
        if exchange_transformed is None:
            exchange_transformed = self.world > 1
        self.exchange_transformed = bool(exchange_transformed and can)
        self.h_full = [None] * self.parities
        # exchange buffers (N > 1): padded send block, gathered blocks, row index that undoes the padding
        self._xchg = None
        if self.world > 1 or self.exchange_transformed:
            counts = [int(c) for c in sharded.counts] if self.world > 1 else [self.local_hi - self.local_lo]
            cmax = max(counts)
            D = first.out_dim + 4 if self.exchange_transformed else base.shape[1]
            if self.exchange_transformed:
                n_all = base.shape[0]
                self.h_full = [torch.zeros((n_all, D), dtype=torch.float32, device=dev) for _ in range(self.parities)]
            src = np.concatenate([r * cmax + np.arange(c, dtype=np.int64) for r, c in enumerate(counts)])
            self._xchg = {"send": [torch.zeros((cmax, D), dtype=torch.float32, device=dev) for _ in range(self.parities)],
                          "recv": [torch.empty((len(counts) * cmax, D), dtype=torch.float32, device=dev)
                                   for _ in range(self.parities)],
                          "rows": torch.as_tensor(src).to(dev), "even": all(c == cmax for c in counts)}
        # Target-row sharding of the interaction level (``shard_upper``; needs ``exchange_transformed``).  Replicated, the
        # graph halves of the interaction layers cost the same ~2 ms on every rank of configs[3] whatever N is, which caps
        # the 1 -> 8 ratio below 6x (profiles/r04_cfg3_emulated_ranks.txt).  Sharded, a rank runs every layer but the last
        # on ITS OWN target rows only (softmax statistics stay replicated: 4 B per edge), applies the next layer's row-wise
        # half to them and all-gathers [N_r, d + 2] again: L - 1 more all-gathers of N x (d + 2) x 4 B per step (102 MB at
        # configs[3]) in exchange for (N - 1) / N of L - 1 gathers; the last layer computes the rows its pairs read from
        # the full table (``prune_last``).  Default: N > 1 and >= 32 768 interaction rows (below that the extra
        # collectives' launch latency is the same size as the compute they save: configs[1]).
        # ``collective=True``: call the all-gather even with ONE rank (a process group must be initialised and the rank must own
        # every interaction row) -- the RCCL call path, its stream ordering against the captured graphs included, on a box
        # with a single GPU (tests/shard_upper_worker.py)
        self.collective = bool(self.world > 1 or collective)
        if self.collective and self.world <= 1:
            if not self.exchange_transformed or self.local_hi - self.local_lo != base.shape[0]:
                raise ValueError("blgnn: collective=True at one rank needs exchange_transformed and every interaction row local")
        chain = self.upper_layers[1:-1]
        can_shard = (self.exchange_transformed and len(chain) >= 2 and
                     all(isinstance(l, NodeEmbedding) and l.type == 'gat' and l.higher_level and not l.normalize
                         for l in chain))
        if shard_upper is None:
            shard_upper = (self.world > 1 and base.shape[0] >= 32768
                           and os.environ.get("BLGNN_ENGINE_SHARD_UPPER", "1") != "0")
        self.shard_upper = bool(shard_upper and can_shard)
        self.h_more, self._xchg_more, self.own_rows = [], [], None
        if self.shard_upper:
            n_all, cmax, nblk = base.shape[0], self._xchg["send"][0].shape[0], self._xchg["recv"][0].shape[0]
            self.own_rows = torch.arange(self.local_lo, self.local_hi, dtype=torch.int64, device=dev)
            for layer in chain[1:]:
                D = layer.out_dim + 4
                self.h_more.append([torch.zeros((n_all, D), dtype=torch.float32, device=dev) for _ in range(self.parities)])
                self._xchg_more.append({"send": [torch.zeros((cmax, D), dtype=torch.float32, device=dev)
                                                 for _ in range(self.parities)],
                                        "recv": [torch.empty((nblk, D), dtype=torch.float32, device=dev)
                                                 for _ in range(self.parities)]})
        self.n_upper_segments = len(chain) if self.shard_upper else 1
        # Pinned-memory copies leave events of torch's host allocator on the stream they ran on; those must
        # never sit on a stream this engine destroys (close()), so the id copies run on an ordinary side
        # stream and the interaction-level stream waits for an event.
        self.stream_c = torch.cuda.Stream(device=dev) if self.overlap else None
        # ``parity_streams``: the two pipeline parities on UNRELATED stream pairs, ordered by nothing but the events of
        # ``step()`` (diagnostic: tests/test_gpu_model.py checks that the parities' graphs touch no common buffer that
        # either writes -- by default both parities' sweeps share stream L and both interaction passes stream U, which
        # would hide such a buffer).  Not a speed-up: measured 0.34 vs 0.20 ms per step at configs[1] in round 3.
        # ``serialize``: sweep and interaction pass of consecutive steps on ONE stream (still double-buffered, still
        # asynchronous to the host, the id copies still on their own stream).  Opt-in (None = False = side by side, see
        # capture()); BLGNN_ENGINE_SERIALIZE = 0 | 1 overrides None for A/B runs.
        if serialize is None and os.environ.get("BLGNN_ENGINE_SERIALIZE") in ("0", "1"):      # (A/B switch)
            serialize = os.environ["BLGNN_ENGINE_SERIALIZE"] == "1"
        self.serialize = serialize
        self.parity_streams = bool(parity_streams and self.overlap)
        self.serialized = False
        self._sl = [self.stream_l, torch.cuda.Stream(device=dev) if self.parity_streams else self.stream_l]
        self._su = [self.stream_u, torch.cuda.Stream(device=dev) if self.parity_streams else self.stream_u]
        self.ev_lower = [None] * self.parities
        self.ev_upper = [None] * self.parities
        self.ev_ids = [None] * self.parities
        self.step_no = 0
        self._stamp = None
        self._watch = None
        self._closed = False

    # ---- weights watched for in-place changes after capture -----------------------------------
    def _weights_stamp(self, relist=False):
        """(sum of versions, storage addresses) of the model's parameters and buffers.  The tensor OBJECTS are listed once
        per capture: walking the module tree on every step cost ~100 us of host time -- at configs[1] the step was bound
        by exactly that (190 us enqueue time for 165 us of GPU work).  In-place updates (optimizers, load_state_dict)
        and ``.data`` swaps are seen; replacing a parameter or module OBJECT after capture needs a new ``capture()``."""
        if relist or self._watch is None:
            self._watch = list(self.model.parameters()) + list(self.model.buffers())
        ts = self._watch
        return (sum([t._version for t in ts]), tuple([t.data_ptr() for t in ts]))

    # ---- pair ids ---------------------------------------------------------------------------------
    def rows_of(self, pair_gids):
        """[B, 2] graph ids -> [2, B] int64 rows of the tensor LinkPred reads (gs_map lookup, vectorised)."""
        g = np.asarray(pair_gids, dtype=np.int64).reshape(-1, 2)
        if g.shape[0] != self.B:
            raise ValueError("blgnn: this engine scores %d pairs per step, got %d" % (self.B, g.shape[0]))
        if self._row_lut is not None:
            # bounds first: a negative id would wrap around under numpy indexing and score another drug's row
            if ((g < 0) | (g >= self._row_lut.size)).any():
                raise KeyError("unknown graph id in pair batch")
            rows = self._row_lut[g]
            if (rows < 0).any():
                raise KeyError("unknown graph id in pair batch")
            return np.ascontiguousarray(rows.T)
        b = BatchData(g, self.template.dataset, is_train=False)
        return np.ascontiguousarray(self.upper_layers[-1].pair_row_ids(b).T)

    def _feed(self, which, parity):
        """Asynchronous H2D copy of one batch's row ids into the parity's static buffer, on the CURRENT
        stream.  ``which``: list index or [B, 2] graph ids."""
        if isinstance(which, (int, np.integer)):
            src = self.host_ids[int(which) % self.nb]
        else:
            if self.ev_ids[parity] is not None:               # the staging buffer's previous copy has run
                self.ev_ids[parity].synchronize()
            src = self.stage_ids[parity]
            # a plain memcpy into the pinned block: Tensor.copy_ on >= 32 768 elements goes through torch's intra-op thread
            # pool, which on a box whose CPU quota is a fraction of its visible cores stalls for 10-70 ms per call (measured
            # at 32 768 pairs per step: the step's GPU work is 12 ms)
            np.copyto(self.stage_np[parity], self.rows_of(which))
        self.ids_dev[parity].copy_(src, non_blocking=True)
        self.ev_ids[parity] = torch.cuda.Event()
        self.ev_ids[parity].record(torch.cuda.current_stream())

    # ---- the three pieces of one forward -------------------------------------------------
    def lower(self, parity=0, readout=True):
        """-> what this rank contributes to the exchange: its pooled rows (a view of init_x: local drugs are a
        contiguous row range), or -- ``exchange_transformed`` -- the send block holding the first interaction
        layer's transform of those rows.  ``readout=False`` stops before the node -> graph readout and keeps
        the activations for ``upper(..., readout=True)``."""
        self.graph.init_x = self.init_x[parity]
        if self.fused is not None:
            self.fused.run(self.model, self.init_x[parity])
        else:
            with torch.no_grad():
                m = self.model
                m.acts = [self.lower_batch.merge_data['merge'].x]
                m._readout_direct = True            # the readout layer runs here or at the head of upper()
                for layer in (self.lower_layers if readout else self.lower_layers[:-1]):
                    m.acts.append(layer(m.acts[-1], self.lower_batch, m))
                if not readout:
                    self.lower_acts[parity] = list(m.acts)
        pooled = self.init_x[parity][self.local_lo:self.local_hi]
        if not self.exchange_transformed:
            return pooled
        first = self.upper_layers[1]
        d = first.out_dim
        send = self._xchg["send"][parity]
        rows = pooled.shape[0]
        _, sc = first.gat_transform_rows(pooled, h=send[:rows, :d])
        send[:rows, d:d + 2].copy_(sc.t())                     # the two scores ride in columns d, d + 1
        return send[:rows]

    def exchange(self, pooled, parity=0, stage=0):
        """The collective of the path: all-gather of the pooled drug rows (padded to the largest
        shard) into persistent buffers, then ONE row gather that drops the padding into init_x.  ``stage`` > 0
        (``shard_upper``): the same for the rows a rank has produced for interaction layer ``stage + 1``."""
        if self.world <= 1 and not self.exchange_transformed:
            return
        x = self._xchg
        if stage > 0:
            more = self._xchg_more[stage - 1]
            send, recv, dest = more["send"][parity], more["recv"][parity], self.h_more[stage - 1][parity]
        else:
            send, recv = x["send"][parity], x["recv"][parity]
            dest = self.h_full[parity] if self.exchange_transformed else self.init_x[parity]
        if not self.exchange_transformed:
            send[:pooled.shape[0]].copy_(pooled)               # (transformed rows were written into ``send``)
        if self.collective:
            import torch.distributed as dist
            dist.all_gather_into_tensor(recv, send, group=self.sharded.group if self.sharded is not None else None)
        else:                                                  # one rank (tests, --emulate-world): its own rows only
            n = self.local_hi - self.local_lo
            dest[self.local_lo:self.local_hi].copy_(send[:n])
            return
        if x["even"]:
            dest.copy_(recv)
        else:
            torch.index_select(recv, 0, x["rows"], out=dest)

    def upper(self, parity=0, readout=False, segment=None):
        """Interaction-level layers + pair scorer on the rows in ``ids_dev[parity]``.  With ``shard_upper`` the pass is
        ``n_upper_segments`` pieces with an all-gather between consecutive ones: ``segment`` = which piece to enqueue
        (what a captured graph holds), None = all of them with their exchanges (eager)."""
        if self.shard_upper:
            if segment is not None:
                return self._upper_segment(parity, segment, readout)
            out = None
            for k in range(self.n_upper_segments):
                out = self._upper_segment(parity, k, readout)
                if k < self.n_upper_segments - 1:
                    self.exchange(None, parity, stage=k + 1)
            return out
        self.graph.init_x = self.init_x[parity]
        with torch.no_grad():
            m, b = self.model, self.template
            if readout:                                        # the sweep's readout, deferred to this graph
                m.acts = list(self.lower_acts[parity])
                self.lower_layers[-1](m.acts[-1], self.lower_batch, m)
            b._pair_row_ids = (self.ids_dev[parity][0], self.ids_dev[parity][1])
            m.acts = [b.merge_data['merge'].x]
            layers = self.upper_layers
            if self.exchange_transformed:
                # LoadInteraction (builds / caches the interaction CSR), then the first GAT layer's graph half on
                # the gathered table [N, d | s_dst s_src]
                m.acts.append(layers[0](m.acts[-1], b, m))
                first, hf = layers[1], self.h_full[parity]
                d = first.out_dim
                sc = hf[:, d:d + 2].t().contiguous()
                m.acts.append(first.gat_forward_from_transform(hf[:, :d], sc, b, m))
                layers = layers[2:]
            prune = self._prune_layer() if self.prune_last else None
            for layer in layers:
                if layer is prune:
                    # the last interaction-level layer feeds only the pair scorer: compute the rows it reads
                    layer._out_rows = self._scored_rows(parity)
                try:
                    m.acts.append(layer(m.acts[-1], b, m))
                finally:
                    if layer is prune:
                        layer._out_rows = None
            return m.acts[-1]

    def _upper_segment(self, parity, k, readout=False):
        """Piece ``k`` of the target-row-sharded interaction pass: the graph half of interaction layer k + 1 on the gathered
        table of its transformed rows -- on this rank's own target rows, followed by the next layer's row-wise half on
        them (written into that stage's send block); the last piece computes the rows the scored pairs read and scores."""
        self.graph.init_x = self.init_x[parity]
        with torch.no_grad():
            m, b = self.model, self.template
            chain, lo, hi = self.upper_layers[1:-1], self.local_lo, self.local_hi
            if k == 0:
                b._pair_row_ids = (self.ids_dev[parity][0], self.ids_dev[parity][1])
                m.acts = [b.merge_data['merge'].x]
                m.acts.append(self.upper_layers[0](m.acts[-1], b, m))       # LoadInteraction (cached CSR)
            layer = chain[k]
            hf = self.h_full[parity] if k == 0 else self.h_more[k - 1][parity]
            d = layer.out_dim
            sc = hf[:, d:d + 2].t().contiguous()
            if k < len(chain) - 1:
                x = layer.gat_forward_from_transform(hf[:, :d], sc, b, m, rows=self.own_rows)
                nxt, send = chain[k + 1], self._xchg_more[k]["send"][parity]
                dn, n = nxt.out_dim, hi - lo
                _, sc2 = nxt.gat_transform_rows(x[lo:hi], h=send[:n, :dn])
                send[:n, dn:dn + 2].copy_(sc2.t())
                return send[:n]
            rows = None
            if self.prune_last and 4 * self.B < self.init_x[0].shape[0]:
                rows = self._scored_rows(parity)
            m.acts.append(layer.gat_forward_from_transform(hf[:, :d], sc, b, m, rows=rows))
            m.acts.append(self.upper_layers[-1](m.acts[-1], b, m))
            return m.acts[-1]

    def _scored_rows(self, parity):
        """The interaction rows this step's pairs read, each ONCE, as a static-shape list (ascending, duplicates replaced
        by -1: blgnn_gat_aggregate skips negative entries).  Pairs are drawn from the interaction EDGES, so a row appears
        in proportion to its degree: at configs[3] the raw [2 B] list of a rank's 8 192 pairs made the pruned last layer
        gather more edges than the whole graph has (632 us against 340 for every row; rocprofv3, emulated N = 8)."""
        return self.dedup_rows(self.ids_dev[parity].view(-1))

    @staticmethod
    def dedup_rows(ids):
        """int64 ids [m] -> [m]: the distinct ids ascending, every repeat replaced by -1 (static shape: capturable)"""
        s = torch.sort(ids).values
        dup = torch.cat([torch.zeros(1, dtype=torch.bool, device=s.device), s[1:] == s[:-1]])
        return torch.where(dup, torch.full_like(s, -1), s)

    def _prune_layer(self):
        """The last interaction-level conv layer when it is a plain GAT layer directly in front of the pair scorer and
        the batch reads fewer than half of the interaction rows (else None: computing every row once is cheaper)."""
        last = self.upper_layers[-2] if len(self.upper_layers) >= 2 else None
        if not (isinstance(last, NodeEmbedding) and last.type == 'gat' and last.higher_level and not last.normalize):
            return None
        if self.exchange_transformed and last is self.upper_layers[1]:
            return None
        if 4 * self.B >= self.init_x[0].shape[0]:
            return None
        return last

    def eager_step(self, which):
        self._feed(which, 0)
        self.exchange(self.lower(0), 0)
        return self.upper(0)

    # ---- capture ------------------------------------------------------------------------------
    def capture(self):
        """Run once eagerly per parity (CSR builds and derived-weight packing must not fall into a
        capture), then record the graphs."""
        for p in range(self.parities):
            self._feed(0, p)
            self.exchange(self.lower(p), p)
            self.upper(p)
        torch.cuda.synchronize()
        self._stamp = self._weights_stamp(relist=True)
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
                gs = []
                for k in range(self.n_upper_segments):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        self.outs[p] = self.upper(p, readout=defer, segment=k if self.shard_upper else None)
                    gs.append(g)
                self.g_upper[p] = gs
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.overlap and not self.parity_streams and self._masked is None:
            ser = self.serialize
            if ser is None:
                # side by side by default.  (First half of round 4 decided this from the two graphs' durations and ran sweep
                # and pass in a row when the sweep was >= 4x the pass: the 27-35 ms "episodes" that motivated it were the host
                # stalling in _feed's staging copy -- see there -- not starvation on the device.  With that fixed, 3-second
                # windows at configs[3]: N = 1 23.04 side by side vs 23.43 in a row, emulated N = 2 11.90 vs 12.30, no episode.)
                ser = False
            self.serialized = bool(ser)
            self._su = [self._sl[0], self._sl[1]] if self.serialized else [self.stream_u, self.stream_u]

    # ---- steady state ---------------------------------------------------------------------------
    def step(self, which):
        """Enqueue one forward (asynchronous); ``which`` = index into the batch list or a [B, 2] array
        of graph ids.  Returns the tensor the pair scores land in (valid once the step has run; it is
        reused two steps later)."""
        if self._closed:
            raise RuntimeError("blgnn: engine is closed")
        if not self.use_graph:
            return self.eager_step(which)
        if self._weights_stamp() != self._stamp:
            # weights / BN statistics changed after capture: the graphs hold pointers to tensors derived
            # from the old values (packed images, folded affines).  Re-derive and re-capture.
            self.synchronize()
            self.capture()
        if not self.overlap:
            self._feed(which, 0)
            self.g_lower[0].replay()
            self.exchange(self.pooled[0], 0)
            self._replay_upper(0)
            return self.outs[0]
        p = self.step_no & 1
        self.step_no += 1
        cur = torch.cuda.current_stream()
        sl, su = self._sl[p], self._su[p]
        with torch.cuda.stream(sl):
            if self.ev_upper[p] is not None:                   # init_x[p] is free once step k-2 scored
                sl.wait_event(self.ev_upper[p])
            else:
                sl.wait_stream(cur)
            self.g_lower[p].replay()
            self.exchange(self.pooled[p], p)
            self.ev_lower[p] = torch.cuda.Event()
            self.ev_lower[p].record(sl)
        with torch.cuda.stream(self.stream_c):
            if self.ev_upper[p] is not None:                   # ids_dev[p] is free once step k-2 scored
                self.stream_c.wait_event(self.ev_upper[p])
            else:
                self.stream_c.wait_stream(cur)
            self._feed(which, p)
        with torch.cuda.stream(su):
            su.wait_event(self.ev_ids[p])
            su.wait_event(self.ev_lower[p])
            self._replay_upper(p)
            self.ev_upper[p] = torch.cuda.Event()
            self.ev_upper[p].record(su)
        return self.outs[p]

    def _replay_upper(self, p):
        """the interaction pass's captured piece(s) on the current stream, ``shard_upper``'s all-gathers between them"""
        gs = self.g_upper[p]
        for k, g in enumerate(gs):
            g.replay()
            if k < len(gs) - 1:
                self.exchange(None, p, stage=k + 1)

    def synchronize(self):
        if self.overlap and self.stream_l is not None:
            cur = torch.cuda.current_stream()
            for st in set(self._sl + self._su + [self.stream_c]):
                cur.wait_stream(st)
        torch.cuda.synchronize()

    def close(self):
        """Drain the device, drop the captured graphs and destroy the CU-masked streams this engine
        owns.  Idempotent; the engine cannot step afterwards."""
        if self._closed:
            return
        self._closed = True
        torch.cuda.synchronize()
        self.g_lower = [None] * self.parities
        self.g_upper = [None] * self.parities
        self.ev_lower = [None] * self.parities
        self.ev_upper = [None] * self.parities
        self.ev_ids = [None] * self.parities
        self.stream_l = self.stream_u = self.stream_c = None
        self._sl = self._su = [None, None]
        if self._masked is not None:
            self._masked.close()
            self._masked = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
