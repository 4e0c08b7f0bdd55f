"""``LoadInteractionGraph`` -- drop-in for model/layers_load_interaction_graph.py.

The reference converts the networkx interaction graph to tensors on EVERY forward
(sorted(list(g.edges)) -> H2D -> to_undirected -> is_undirected assert).  The graph
is static per fold, so here the symmetric sorted edge_index and its CSR are built
once per graph object and only ``init_x`` is re-read.
"""

import torch.nn as nn

from .config import FLAGS
from .graph import GraphData
from .layers_util import convert_nx_to_pyg_graph, create_edge_index

__all__ = ["LoadInteractionGraph"]


class LoadInteractionGraph(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x, batch_data, model):
        assert hasattr(batch_data, "interaction_combo_nxgraph")
        batch_data.merge_higher_level['merge'] = \
            convert_nx_to_pyg_graph(batch_data.interaction_combo_nxgraph)
        if FLAGS.different_edge_type_aggr:
            edges = batch_data.merge_higher_level['edges'] = {}
            for k, v in batch_data.dataset.interaction_nxgraphs.items():
                data = getattr(v, "_blgnn_edge_data", None)
                if data is None:
                    data = GraphData(edge_index=create_edge_index(v)[0],
                                     num_nodes=v.number_of_nodes())
                    data.symmetric_by_construction = True        # create_edge_index: to_undirected
                    v._blgnn_edge_data = data
                edges[k] = data
        return batch_data.merge_higher_level['merge'].x
