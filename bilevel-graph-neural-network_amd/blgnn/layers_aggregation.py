"""Node -> graph readout: drop-ins for the reference's model/layers_aggregation.py.

``scatter_mean`` / ``scatter_add`` over the sorted batch vector become one
segment-reduce kernel per pooled activation, each writing straight into its
column slice of the concatenated output; with higher-level layers present the
per-gid ``init_x`` row writes (layers_aggregation.py:68-74, a python loop of G
indexed assignments) become one row-scatter kernel.
"""

import numpy as np
import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .config import FLAGS
from .layers_util import MLP

__all__ = ["NodeAggregationPairs", "NodeAggregation", "DeepSets", "GMNAggregatorPairs"]


def _segments(batch_data, pair_batch):
    if pair_batch and hasattr(batch_data, "pair_batch"):
        b = batch_data.pair_batch
        return ops.segments_from_batch(b, int(batch_data.num_pairs))
    return batch_data.merge_data['merge'].segments()


class _SegmentPool(object):
    def __init__(self, kind):
        self.kind = kind

    def __call__(self, x, seg, out=None):
        return ops.segment_reduce(seg, x, self.kind, out=out)


class NodeAggregationPairs(nn.Module):
    def __init__(self, style, concat_multi_scale=False, in_dim=None, out_dim=None,
                 num_mlp_layers=None):
        super().__init__()
        self.style = style
        self.concat_multi_scale = concat_multi_scale
        if self.style == "avg_pool":
            self.agg_func = _SegmentPool("mean")
        elif self.style == "sum":
            self.agg_func = _SegmentPool("sum")
        elif self.style == 'deepsets':
            self.agg_func = DeepSets(in_dim, out_dim, num_mlp_layers)
        elif self.style == 'gmn_aggr':
            self.agg_func = GMNAggregatorPairs(in_dim, out_dim, bn=True)
        else:
            raise NotImplementedError("{} is not implemented".format(style))

    def forward(self, ins, batch_data, model, pair_batch=False, out=None):
        ops.require_device(ins, "ins")
        seg = _segments(batch_data, pair_batch)
        if ag.training_active(self) and ins.requires_grad:
            if out is not None:
                raise NotImplementedError("blgnn: train mode does not write into a caller-provided slice")
            if not isinstance(self.agg_func, _SegmentPool):
                # 'deepsets' / 'gmn_aggr' (model/layers_aggregation.py:45-56,78-95): MLPs and reductions as autograd nodes
                if self.concat_multi_scale:
                    xs = model.acts[1:] if 'gmn' not in FLAGS.model else \
                        [x for i, x in enumerate(model.acts[1:]) if i % 2 == 1]
                    return torch.cat([self.agg_func(x, seg) for x in xs], dim=1)     # (placement: data movement only)
                return self.agg_func(ins, seg)
            if self.concat_multi_scale:
                xs = model.acts[1:] if 'gmn' not in FLAGS.model else \
                    [x for i, x in enumerate(model.acts[1:]) if i % 2 == 1]
            else:
                xs = [ins]
            return ag.segment_reduce_multi(seg, xs, self.agg_func.kind)
        plan = getattr(model, "_readout_plan", None)
        if plan is not None and plan.agg is self and plan.batch is batch_data and not pair_batch:
            model._readout_plan = None
            if plan.complete():
                # the pooled rows were accumulated by the producing layers' epilogues (layers.ReadoutPlan)
                if out is None or out.data_ptr() == plan.dest.data_ptr():
                    return plan.dest
                out.copy_(plan.dest)
                return out
        with torch.no_grad():
            if self.concat_multi_scale:
                if 'gmn' not in FLAGS.model:
                    xs = model.acts[1:]
                else:
                    xs = [x for i, x in enumerate(model.acts[1:]) if i % 2 == 1]
                if any(getattr(x, "_blgnn_pooled", None) is not None or getattr(x, "_blgnn_pool_of", None) is not None
                       for x in xs):
                    out = self._concat_with_pooled(xs, seg, out)
                elif isinstance(self.agg_func, _SegmentPool) and len(xs) <= 8:
                    out = ops.segment_reduce_multi(seg, xs, self.agg_func.kind, out=out)
                elif isinstance(self.agg_func, _SegmentPool):
                    G = seg.numel() - 1
                    widths = [x.shape[1] for x in xs]
                    out = torch.empty((G, sum(widths)), dtype=torch.float32, device=ins.device)
                    c = 0
                    for x, w in zip(xs, widths):
                        self.agg_func(x, seg, out=out[:, c:c + w])
                        c += w
                else:
                    out = torch.cat([self.agg_func(x, seg) for x in xs], dim=1)
            elif getattr(ins, "_blgnn_pooled", None) is not None:
                out = self._concat_with_pooled([ins], seg, out)
            else:
                out = self.agg_func(ins, seg)
        return out

    def _concat_with_pooled(self, xs, seg, out):
        """Readout when the last propagator already pooled its rows (layers_gmn.py ``_readout_of``: its [G, D] result
        carries ``_blgnn_pooled``): node-row inputs are reduced in runs, pooled inputs copied into their columns."""
        kind = self.agg_func.kind
        G = seg.numel() - 1
        widths = [x.shape[1] for x in xs]
        if out is None:
            out = torch.empty((G, sum(widths)), dtype=torch.float32, device=xs[0].device)
        def pooled_of(x):
            """-> (kind, [G, D]) when ``x`` is already pooled (``_blgnn_pooled``) or brings its pooled rows along
            (``_blgnn_pool_of``: node rows whose producing GEMM pooled them in its epilogue), else None"""
            tag = getattr(x, "_blgnn_pooled", None)
            return (tag, x) if tag is not None else getattr(x, "_blgnn_pool_of", None)

        c = k = 0
        while k < len(xs):
            have = pooled_of(xs[k])
            if have is not None:
                if have[0] != kind or have[1].shape[0] != G:
                    raise ValueError("blgnn: pooled layer output does not match this readout")
                out[:, c:c + widths[k]].copy_(have[1])                      # (placement: data movement only)
                c += widths[k]
                k += 1
                continue
            run = k
            while run < len(xs) and run - k < 8 and pooled_of(xs[run]) is None:
                run += 1
            w = sum(widths[k:run])
            ops.segment_reduce_multi(seg, xs[k:run], kind, out=out[:, c:c + w])
            c += w
            k = run
        return out


class DeepSets(nn.Module):
    """layers_aggregation.py:45-56."""

    def __init__(self, in_dim, out_dim, num_mlp_layers):
        super(DeepSets, self).__init__()
        self.phi = MLP(in_dim, out_dim, num_hidden_lyr=num_mlp_layers - 1)
        self.rho = MLP(out_dim, out_dim, num_hidden_lyr=num_mlp_layers - 1)

    def forward(self, ins, seg, out=None):
        if ag.training_active(self) and ins.requires_grad:
            return self.rho(ag.segment_reduce_multi(seg, [self.phi(ins)], "mean"))
        return self.rho(ops.segment_reduce(seg, self.phi(ins), "mean"), out=out)


class NodeAggregation(NodeAggregationPairs):
    """aggregates node embeddings for a graph embedding (one unique embedding per graph)"""

    def __init__(self, style, is_last_layer, **kwargs):
        super(NodeAggregation, self).__init__(style, **kwargs)
        self.is_last_layer = is_last_layer

    def readout_destination(self, batch_data, width):
        """The slice of the interaction graph's init_x this batch's pooled rows go to, when the batch graphs are a
        contiguous run of interaction rows (else None)."""
        if not FLAGS.higher_level_layers:
            return None
        init_x = batch_data.interaction_combo_nxgraph.init_x
        rng = batch_data.interaction_row_range() if hasattr(batch_data, "interaction_row_range") else None
        if rng is None or not isinstance(init_x, torch.Tensor) or init_x.shape[1] != width:
            return None
        return init_x[rng[0]:rng[1]]

    def forward(self, x, batch_data, model, pair_batch=False):
        if not FLAGS.higher_level_layers:
            return super().forward(x, batch_data, model, pair_batch)
        if ag.training_active(self) and x.requires_grad:
            # training (src/train.py:48-72 runs the sweep WITH gradients): the pooled rows enter init_x as an
            # autograd value, so the loss reaches the molecular layers through the interaction-level layers
            out = super().forward(x, batch_data, model, pair_batch)
            graph = batch_data.interaction_combo_nxgraph
            rows = batch_data.interaction_rows_of_batch_graphs()
            rng = batch_data.interaction_row_range()
            n_all = graph.number_of_nodes()
            if rng is not None and rng[0] == 0 and rng[1] == n_all:
                graph.init_x = out                                  # the sweep covers every drug, in gs_map order
            else:
                # a chunk of the sweep (src/train.py:48-72 always sweeps in several chunks and writes
                # init_x[gs_map[gid]] = out[ind] in place): the rows earlier chunks wrote keep their autograd history --
                # get_initial_embd detaches init_x ONCE, before the first chunk of a sweep
                base = graph.init_x if isinstance(graph.init_x, torch.Tensor) and \
                    graph.init_x.shape[1] == out.shape[1] else \
                    torch.zeros((n_all, out.shape[1]), dtype=torch.float32, device=out.device)
                if rng is not None and rng[0] == 0 and base.requires_grad:
                    # the first chunk of a sweep driven WITHOUT get_initial_embd (a loop calling model(b) per chunk):
                    # start from values, not from the previous iteration's (freed) autograd graph
                    base = base.detach()
                graph.init_x = base.index_copy(0, rows.long(), out)     # row placement: data movement only
            return out
        # layers_aggregation.py:68-74: init_x[gs_map[gid]] = out[ind]
        graph = batch_data.interaction_combo_nxgraph
        init_x = graph.init_x
        rng = batch_data.interaction_row_range() if hasattr(batch_data, "interaction_row_range") else None
        if rng is not None and isinstance(init_x, torch.Tensor) and self.concat_multi_scale and \
                isinstance(self.agg_func, _SegmentPool) and init_x.shape[1] == self._pooled_width(model):
            # the batch graphs are a contiguous run of interaction rows: the readout writes that slice
            # of init_x directly and the returned tensor is a view of it
            return super().forward(x, batch_data, model, pair_batch, out=init_x[rng[0]:rng[1]])
        out = super().forward(x, batch_data, model, pair_batch)
        rows = batch_data.interaction_rows_of_batch_graphs()
        if not isinstance(init_x, torch.Tensor) or init_x.shape[1] != out.shape[1]:
            init_x = torch.zeros((graph.number_of_nodes(), out.shape[1]), dtype=torch.float32,
                                 device=out.device)
            graph.init_x = init_x
        ops.scatter_rows(out, rows, init_x)
        return out

    def _pooled_width(self, model):
        if 'gmn' not in FLAGS.model:
            xs = model.acts[1:]
        else:
            xs = [x for i, x in enumerate(model.acts[1:]) if i % 2 == 1]
        return sum(x.shape[1] for x in xs) if len(xs) <= 8 else -1


class GMNAggregatorPairs(nn.Module):
    """layers_aggregation.py:78-95."""

    def __init__(self, input_dim, output_dim, bn):
        super().__init__()
        self.out_dim = output_dim
        self.sigmoid = nn.Sigmoid()
        self.weight_func = MLP(input_dim, output_dim, num_hidden_lyr=1,
                               hidden_channels=[output_dim], bn=bn)
        self.gate_func = MLP(input_dim, output_dim, num_hidden_lyr=1,
                             hidden_channels=[output_dim], bn=bn)
        self.mlp_graph = MLP(output_dim, output_dim, num_hidden_lyr=1,
                             hidden_channels=[output_dim], bn=bn)

    def forward(self, x, seg, out=None):
        if ag.training_active(self) and x.requires_grad:
            # training: the three MLPs (train-mode BatchNorm inside), the gate product and the sum as autograd nodes
            gated_x = self.gate_func(x, final_act="sigmoid")
            return self.mlp_graph(ag.segment_reduce_multi(seg, [ag.hadamard(gated_x, self.weight_func(x))], "sum"))
        weighted_x = self.weight_func(x)
        gated_x = self.gate_func(x, final_act="sigmoid")                # the sigmoid rides on the last GEMM's epilogue
        graph_embeddings = ops.segment_reduce(seg, ops.hadamard(gated_x, weighted_x), "sum")
        return self.mlp_graph(graph_embeddings, out=out)
