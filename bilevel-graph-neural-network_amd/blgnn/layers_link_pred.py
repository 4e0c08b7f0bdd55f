"""``LinkPred`` -- drop-in for model/layers_link_pred.py (the pair scorer that
consumes the path's output).  normalize + gather + concat is one kernel, the MLP
runs on the fp32 matrix cores, and the B per-pair ``.item()`` device syncs of
layers_link_pred.py:69-70 / utils/data/graph.py:35-39 are replaced by one tensor
(``batch_data.pair_preds``); ``pair.get_link_pred()`` reads it lazily.
"""

import numpy as np
import torch
import torch.nn as nn

from . import autograd as ag
from . import ops
from .config import FLAGS
from .layers_util import MLP, act_name

__all__ = ["LinkPred"]


class LinkPred(nn.Module):
    def __init__(self, type, mlp_dim, num_labels, weight_dim=None, batch_unique_graphs=True,
                 multi_label_pred=False):
        super().__init__()
        self.type = type
        self.num_labels = num_labels
        self.batch_unique_graphs = batch_unique_graphs
        self.weight_dim = weight_dim
        self.multi_label_pred = multi_label_pred
        if weight_dim:
            self.weight_matrix = nn.Parameter(torch.zeros((weight_dim, weight_dim)))
            torch.nn.init.xavier_normal_(self.weight_matrix, gain=nn.init.calculate_gain('relu'))
        if self.type not in ["dot_product", "mlp_concat"]:
            raise NotImplementedError
        if self.multi_label_pred:
            assert self.type == 'ntn' or self.type == 'mlp_concat'
        if self.type == 'mlp_concat':
            out = num_labels if multi_label_pred else 1
            dims = self._calc_mlp_dims(mlp_dim * 2, out, division=8)
            self.mlp_concat = MLP(mlp_dim * 2, out, num_hidden_lyr=len(dims),
                                  hidden_channels=dims, bn=False)

    @staticmethod
    def _calc_mlp_dims(mlp_dim, output_dim=1, division=2):
        dim, dims = mlp_dim, []
        while dim > output_dim:
            dim = dim // division
            dims.append(dim)
        return dims[:-1]

    def pair_row_ids(self, batch_data):
        """layers_link_pred.py:46-54: which rows of ``ins`` each pair reads."""
        batch_gids = batch_data.batch_gids
        if isinstance(batch_gids, torch.Tensor):
            batch_gids = batch_gids.cpu().numpy()
        if self.batch_unique_graphs:
            if FLAGS.higher_level_layers and hasattr(batch_data.dataset, "rows_of_gids"):
                ids = batch_data.dataset.rows_of_gids(batch_gids)            # gs_map through its dense table
            else:
                table = batch_data.dataset.gs_map if FLAGS.higher_level_layers \
                    else batch_data.merge_data['gids_to_batch_ind']
                ids = np.vectorize(table.get)(batch_gids)
        else:
            ids = np.arange(batch_gids.shape[0] * batch_gids.shape[1]).reshape(batch_gids.shape)
        return np.ascontiguousarray(ids, dtype=np.int64)

    def forward(self, ins, batch_data, model):
        ops.require_device(ins, "ins")
        ids = getattr(batch_data, "_pair_row_ids", None)
        if ids is None:
            if not self.batch_unique_graphs:
                # pair mode (layers_link_pred.py:53-54: ids = arange over the batch's graphs, pair i = graphs 2 i, 2 i + 1):
                # built on the device -- a host-to-device copy HERE, at the end of the forward, makes the host wait for
                # every kernel the forward has enqueued (16 ms per fresh batch at configs[2], measured)
                n2 = 2 * int(len(batch_data.batch_gids))
                ids = (torch.arange(0, n2, 2, dtype=torch.int64, device=ins.device),
                       torch.arange(1, n2, 2, dtype=torch.int64, device=ins.device))
            else:
                host = self.pair_row_ids(batch_data)
                ids = (torch.as_tensor(np.ascontiguousarray(host[:, 0])).to(ins.device),
                       torch.as_tensor(np.ascontiguousarray(host[:, 1])).to(ins.device))
            batch_data._pair_row_ids = ids
        ids1, ids2 = ids
        mlp = getattr(self, "mlp_concat", None)
        if ag.training_active(self) and ins.requires_grad:
            cat = ag.pair_gather(ag.l2_normalize(ins), ids1, ids2)
            if self.type == "mlp_concat":
                pair_preds = self.mlp_concat(cat, final_act="identity" if self.multi_label_pred else "sigmoid")
            else:
                # 'dot_product' (model/layers_link_pred.py:65-66): sigmoid(<z1, z2>) = product node + a [1, d] row of
                # ones as the Linear's (constant) weight, the sigmoid on its epilogue
                d = ins.shape[1]
                ones = torch.ones((1, d), dtype=torch.float32, device=ins.device)
                pair_preds = ag.linear(ag.hadamard(cat[:, :d], cat[:, d:]), ones, None, "sigmoid").view(-1)
            batch_data.pair_preds = pair_preds
            return pair_preds
        with torch.no_grad():
            # fused head for the interaction-level widths; wider graph embeddings (multi-scale readout of a
            # lower-level-only model at d=256: 768 per graph) go through gather + the matrix-core GEMMs: the
            # head's wide variant (one pair per wave, weights in LDS) took 1.55 ms for 8 192 pairs there
            if (self.type == "mlp_concat" and len(mlp.layers) <= 4 and ins.shape[1] <= 256
                    and max(mlp.layer_channels) <= 2048 and act_name(mlp.activation) != "prelu"):
                # fused head: gather + normalise + concat + MLP (+ sigmoid) in one launch
                pair_preds = ops.link_pred_head(
                    ins, ids1, ids2, [mlp.weight_kn(i) for i in range(len(mlp.layers))],
                    [lin.bias for lin in mlp.layers], hidden_act=act_name(mlp.activation),
                    final_act="identity" if self.multi_label_pred else "sigmoid")
                batch_data.pair_preds = pair_preds
                return pair_preds
            cat = ops.pair_gather_normalize(ins, ids1, ids2)
            if self.type == "mlp_concat":
                pair_preds = self.mlp_concat(
                    cat, final_act="identity" if self.multi_label_pred else "sigmoid")
            else:
                # 'dot_product' (non-default): sigmoid(<z1, z2>) as product kernel + a [d, 1] GEMM of ones with the
                # sigmoid on its epilogue
                d = ins.shape[1]
                ones = torch.ones((d, 1), dtype=torch.float32, device=ins.device)
                pair_preds = ops.linear(ops.hadamard(cat[:, :d], cat[:, d:]), ones, act="sigmoid", transform="f32").view(-1)
        batch_data.pair_preds = pair_preds
        return pair_preds
