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

import torch

__all__ = ["BiLevelEngine"]


class BiLevelEngine(object):
    def __init__(self, model, lower_batch, upper_batches, graph, sharded=None, world_size=1,
                 use_graph=True, overlap=True):
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
        self.stream_l = torch.cuda.Stream(device=dev) if self.overlap else None
        self.stream_u = torch.cuda.Stream(device=dev) if self.overlap else None
        self.ev_lower = [None] * self.parities
        self.ev_upper = [None] * self.parities
        self.step_no = 0

    # ---- the three pieces of one forward -------------------------------------------------
    def lower(self, parity=0):
        self.graph.init_x = self.init_x[parity]
        with torch.no_grad():
            m = self.model
            m.acts = [self.lower_batch.merge_data['merge'].x]
            for layer in self.lower_layers:
                m.acts.append(layer(m.acts[-1], self.lower_batch, m))
            return m.acts[-1]                                  # pooled rows of this rank's drugs

    def exchange(self, pooled, parity=0):
        if self.world > 1:                                     # the one collective of the path
            self.sharded.gather_init_x(pooled, out=self.init_x[parity])

    def upper(self, i, parity=0):
        self.graph.init_x = self.init_x[parity]
        with torch.no_grad():
            m, b = self.model, self.upper_batches[i % self.nb]
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
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    self.pooled[p] = self.lower(p)
                self.g_lower[p] = g
                for i in range(self.nb):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side):
                        self.outs[p][i] = self.upper(i, p)
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
