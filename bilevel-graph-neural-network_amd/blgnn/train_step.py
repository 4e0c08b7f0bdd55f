"""``TrainStep`` -- one training iteration of the bi-level model as the reference's loop runs it
(src/train.py:160-182: molecular sweep with gradients -> init_x -> pair batch -> loss -> backward -> optimizer step),
captured ONCE in a HIP graph and replayed per batch.

Why: after the kernels of the training path were given the parallelism their shapes have, an iteration at the DrugBank
shape is ~350 launches of ~10 us kernels -- 3.6 ms of GPU time behind ~7.7 ms of ctypes / autograd dispatch.  The molecule
set, the interaction graph and every shape of the iteration are static; what changes per batch is which rows the pair
scorer reads and their labels: 2 x 8 B + 4 B per pair, copied into static device buffers in front of the replay.

Everything the captured region executes is the SAME code the eager step executes (the modules' train-mode forwards, the
autograd nodes of blgnn/autograd.py, torch's optimizer): capture records the launches, it does not change them.
``capture=False`` runs the identical step eagerly (tests compare the two).

Scope: one GPU, the bi-level models (lower sweep + interaction level), optimizers whose ``step()`` launches no host-side
decision (SGD, and torch's ``capturable`` variants).  Data-parallel training keeps the eager iteration: the gradient and
BatchNorm-statistics all-reduces of ``autograd.set_batch_norm_sync`` are driven from the host between kernels.
"""

import numpy as np
import torch

from .config import FLAGS
from .forward import get_initial_embd

__all__ = ["TrainStep"]


class TrainStep(object):
    def __init__(self, model, data, optimizer, first_batch, lower_cache=None, capture=True, warmup=3):
        """``first_batch``: a ``BatchData`` (train mode, negatives sampled) of the size every later batch will have; with
        ``capture`` the model trains on it during the ``warmup`` eager iterations in front of the capture (allocator pools,
        optimizer state, cached layouts must exist before recording; the capture itself executes nothing).
        ``optimizer``: a torch optimizer over the model's parameters (SGD and the other capturable ones)."""
        if not (FLAGS.lower_level_layers and FLAGS.higher_level_layers):
            raise NotImplementedError("blgnn: TrainStep drives the bi-level model (lower sweep + interaction level)")
        self.model, self.data, self.opt = model, data, optimizer
        self.lower_cache = lower_cache if lower_cache is not None else []
        self.device = torch.device(FLAGS.device)
        self.num_pairs = int(len(first_batch.batch_gids))
        self.template = first_batch
        dev = self.device
        self.ids = torch.zeros((2, self.num_pairs), dtype=torch.int64, device=dev)
        self.labels = torch.zeros(self.num_pairs, dtype=torch.float32, device=dev)
        self._host_ids = torch.zeros((2, self.num_pairs), dtype=torch.int64).pin_memory()
        self._host_labels = torch.zeros(self.num_pairs, dtype=torch.float32).pin_memory()
        # the template batch reads the static buffers (LinkPred: ``_pair_row_ids``; Loss: ``true_labels``)
        self.template._pair_row_ids = (self.ids[0], self.ids[1])
        self.template.true_labels = lambda: self.labels
        self.scorer = [l for l in model.layers if hasattr(l, "pair_row_ids")][-1]
        self._copied = torch.cuda.Event()
        self._versioned = None
        self.graph = None
        self.loss = None
        self.stream = torch.cuda.Stream(device=dev) if capture else None
        self._fill(first_batch)
        if capture:
            torch.cuda.synchronize(dev)
            with torch.cuda.stream(self.stream):
                for _ in range(max(int(warmup), 1)):             # allocator pools, optimizer state, cached layouts
                    self._iteration()
                self.stream.synchronize()
                self.opt.zero_grad(set_to_none=True)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.loss = self._iteration(zero=False)
            torch.cuda.current_stream(dev).wait_stream(self.stream)

    # ---- one iteration, eager (this is what gets captured) ----------------------------------------------------------
    def _iteration(self, zero=True):
        if zero:
            self.opt.zero_grad(set_to_none=True)
        get_initial_embd(self.data, self.model, cached_batches=self.lower_cache)     # sweep WITH gradients -> init_x
        self.data.dataset.init_interaction_graph_embds(self.device)
        self.model.use_layers = 'higher_layers'
        loss = self.model(self.template)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def _fill(self, batch):
        """rows of the interaction graph each pair reads + labels -> the static buffers (one pinned staging copy each)"""
        if len(batch.batch_gids) != self.num_pairs:
            raise ValueError("blgnn: TrainStep was built for %d pairs per batch, got %d"
                             % (self.num_pairs, len(batch.batch_gids)))
        host = self.scorer.pair_row_ids(batch)
        self._copied.synchronize()               # the previous batch's H2D copies have read the pinned staging buffers
        # (plain memcpys into the pinned blocks: Tensor.copy_ on large CPU tensors goes through torch's intra-op thread pool,
        # which stalls for tens of ms on a box whose CPU quota is a fraction of its visible cores -- blgnn/engine.py _feed)
        np.copyto(self._host_ids.numpy(), host.T)
        np.copyto(self._host_labels.numpy(), np.asarray(batch._labels, dtype=np.float32))
        self.ids.copy_(self._host_ids, non_blocking=True)
        self.labels.copy_(self._host_labels, non_blocking=True)
        self._copied.record(torch.cuda.current_stream(self.device))

    def _bump_versions(self):
        """A replay writes the parameters (optimizer step) and the BatchNorm statistics through the captured kernels:
        ``Tensor._version`` does not move, and every derived-tensor cache of the package (transposed / packed weights, the
        eval-BN fold, the GMN composed weights, the engine's capture stamp) is keyed on (data_ptr, _version).  Host-side
        counter bumps, no launch: an eval forward between two replays (src/train.py:152 validates every
        ``iters_per_validation`` iterations) then re-derives from the weights the replay left."""
        if self._versioned is None:
            seen, ts = set(), []
            for group in self.opt.param_groups:
                for p in group["params"]:
                    if id(p) not in seen:
                        seen.add(id(p)); ts.append(p)
            for t in list(self.model.parameters()) + list(self.model.buffers()):
                if id(t) not in seen:
                    seen.add(id(t)); ts.append(t)
            self._versioned = ts
        torch.autograd.graph.increment_version(self._versioned)

    def step(self, batch):
        """Train on ``batch`` (same number of pairs as the first one).  -> the loss (a device scalar; with capture it is a
        static tensor overwritten by the next call)."""
        if self.graph is None:
            self._fill(batch)
            return self._iteration()
        with torch.cuda.stream(self.stream):
            # the previous replay has been enqueued on this stream: the copies are ordered behind it, and the pinned
            # staging buffers must not be rewritten before the previous copies ran
            self.stream.synchronize()
            self._fill(batch)
            self.graph.replay()
        self._bump_versions()
        # whoever reads the loss (or the parameters) on the caller's stream reads them behind this replay
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return self.loss
