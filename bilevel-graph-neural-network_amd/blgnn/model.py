"""``Model`` -- drop-in for the reference's model/model.py: builds the layer list from
FLAGS, slices it into init / lower / higher level groups (model.py:20-27) and runs
the selected slice sequentially, keeping every activation in ``acts`` because the
multi-scale readout needs them (model.py:34-62).
"""

import torch.nn as nn

from .config import FLAGS
from .layers_factory import create_layers

__all__ = ["Model"]

_SLICES = ('all', 'init_layers', 'lower_layers', 'higher_layers', 'higher_no_eval_layers')


class Model(nn.Module):
    def __init__(self, data, config_layer_type='layer'):
        super(Model, self).__init__()
        self.train_data = data
        self.interaction_num_node_feat = data.dataset.interaction_num_node_feat
        self.num_node_feat = data.num_node_feat
        self.num_hyper_edge_feat = data.num_hyper_edge_feat
        self.num_labels = data.dataset.num_labels
        self.layers = create_layers(self, config_layer_type,
                                    vars(FLAGS)["{}_num".format(config_layer_type)])
        assert len(self.layers) > 0
        self.pred_layer = self.layers[-2]          # pred layer is second to last
        self._use_layers = 'all'
        # the extra attributes re-register the same modules, so a state_dict carries the
        # reference's duplicated prefixes (model.py:18-27) and loads either way
        if FLAGS.lower_level_layers and FLAGS.higher_level_layers:
            self._use_layers = 'init_model'
            k = FLAGS.last_lower_lyr_num
            self.init_layers = self.layers[:k]
            self.lower_layers = self.layers[:k + 1]
            self.higher_level_layers = self.layers[k + 1:]
        elif FLAGS.lower_level_layers:
            self.init_layers = self.layers[:-2]
            self.lower_layers = self.layers[:-2]
        self.layer_output = {}
        self.acts = None

    def _selected(self):
        u = self._use_layers
        if FLAGS.lower_level_layers and FLAGS.higher_level_layers:
            table = {'init_layers': lambda: self.init_layers,
                     'lower_layers': lambda: self.lower_layers,
                     'higher_layers': lambda: self.higher_level_layers,
                     'higher_no_eval_layers': lambda: self.layers[FLAGS.last_lower_lyr_num + 1:-2]}
            if u not in table:
                raise UnboundLocalError("use_layers={!r} selects no layer slice".format(u))
            return table[u]()
        if u == 'higher_no_eval_layers':
            return self.layers[:-2]
        if u == 'lower_layers':
            return self.lower_layers
        if u == 'init_layers':
            return self.init_layers
        return self.layers

    def forward(self, batch_data):
        self.acts = [batch_data.merge_data['merge'].x]
        selected = self._selected()
        # the fused readout may write straight into init_x only when the readout layer is part of this slice
        self._readout_direct = any(hasattr(l, "readout_destination") for l in selected)
        for layer in selected:
            self.acts.append(layer(self.acts[-1], batch_data, self))
        return self.acts[-1]

    def store_layer_output(self, layer, output):
        self.layer_output[layer] = output

    def get_layer_output(self, layer):
        return self.layer_output[layer]

    @property
    def use_layers(self):
        return self._use_layers

    @use_layers.setter
    def use_layers(self, setting):
        assert setting in _SLICES
        self._use_layers = setting
