"""``MetaLayerWrapper`` with ``node_model='gat_multi_edge_aggr'`` (the DrugCombo
upper level) or ``'gcn_multi_edge_aggr'`` -- drop-in for the reachable part of model/layers_meta.py.

out = act( sum_k GATConv_k(x, edge_index_k) ) (or GCNConv_k): each edge type accumulates into the
same output buffer inside the aggregation kernel's epilogue; the activation is
applied by the last edge type's pass.  ``EdgeModel`` / ``NodeModel`` / ``*ConvEdge``
are only reachable with ``use_interaction_edge_attrs=True``, which src/config.py:77-83
forces off -- they raise NotImplementedError here.
"""

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .config import FLAGS
from .layers import GATConv, GCNConv
from .layers_util import create_act, act_name

__all__ = ["MetaLayerWrapper", "NodeModelAggrByEdge"]


class NodeModelAggrByEdge(nn.Module):
    def __init__(self, type, input_dim, num_edge_types, output_dim):
        super().__init__()
        self.type = type
        self.input_dim = input_dim
        self.output_dim = output_dim
        if type == 'gcn':
            self.GNNS = nn.ModuleList([GCNConv(input_dim, output_dim) for _ in range(num_edge_types - 1)])
        elif type == 'gat':
            self.GNNS = nn.ModuleList([GATConv(input_dim, output_dim) for _ in range(num_edge_types - 1)])
        else:
            raise ValueError

    def forward(self, x, edge_index, edge_attr, u, final_act="identity"):
        batch_data = u
        graphs = list(batch_data.merge_higher_level['edges'].values())
        out = None
        for i, g in enumerate(graphs):
            last = i == len(graphs) - 1
            act = final_act if last else "identity"
            if self.type == 'gat':
                c = self.GNNS[i]
                out = ops.gat_conv(g.csr(), x, c.weight, c.att, c.bias,
                                   group=getattr(FLAGS, "gat_softmax_group", "src"),
                                   slope=c.negative_slope, act=act, out=out, accumulate=i > 0)
            else:                                            # 'gcn': model/layers_meta.py:68,74-79
                c = self.GNNS[i]
                out = ops.gcn_conv(g.csr(), x, c.weight, c.bias, act=act, out=out, accumulate=i > 0)
        return out


class _MetaLayer(nn.Module):
    """Holder that keeps the reference's parameter prefix ``meta_layer.node_model.*``."""

    def __init__(self, edge_model, node_model):
        super().__init__()
        self.edge_model = edge_model
        self.node_model = node_model


class MetaLayerWrapper(nn.Module):
    def __init__(self, input_dim, edge_dim, output_dim, edge_model, node_model, act,
                 num_edge_types, higher_level=True):
        super().__init__()
        self.higher_level = higher_level
        self.activation = create_act(act)
        self.num_edge_types = num_edge_types
        if edge_model == 'none':
            self.edge_model = None
        else:
            raise NotImplementedError
        if 'multi_edge_aggr' in node_model:
            assert self.edge_model is None
            assert higher_level
            self.node_model = NodeModelAggrByEdge(node_model.split('_')[0], input_dim,
                                                  num_edge_types, output_dim)
        else:
            raise NotImplementedError(
                "node_model={} needs interaction edge attributes, which src/config.py:77-83 "
                "disables; outside the hot path".format(node_model))
        self.meta_layer = _MetaLayer(self.edge_model, self.node_model)

    def forward(self, ins, batch_data, model):
        ops.require_device(ins, "ins")
        merge = batch_data.merge_higher_level['merge'] if self.higher_level \
            else batch_data.merge_data['merge']
        if ag.training_active(self) and self.node_model.type == 'gat':
            # training: sum over edge types of GATConv + act as one autograd node (blgnn/autograd.py)
            graphs = list(batch_data.merge_higher_level['edges'].values())
            return ag.multi_edge_gat(ins, list(self.node_model.GNNS)[:len(graphs)], [g.csr() for g in graphs],
                                     group=getattr(FLAGS, "gat_softmax_group", "src"),
                                     slope=self.node_model.GNNS[0].negative_slope, act=act_name(self.activation))
        if ag.training_active(self):
            # 'gcn_multi_edge_aggr' (model/layers_meta.py:68,74-79): sum over edge types of GCNConv + act as one node
            graphs = list(batch_data.merge_higher_level['edges'].values())
            return ag.multi_edge_gcn(ins, list(self.node_model.GNNS)[:len(graphs)], [g.csr() for g in graphs],
                                     act=act_name(self.activation))
        with torch.no_grad():
            return self.node_model(ins, merge.edge_index, merge.edge_attr, batch_data,
                                   final_act=act_name(self.activation))
