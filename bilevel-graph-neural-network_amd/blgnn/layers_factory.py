"""Spec-string -> module factory: drop-in for model/layers_factory.py.

Same grammar (``Name:key=val,key=val`` read from ``FLAGS.layer_<i>``), same
registry names (layers_factory.py:179-190), same arity checks and errors.
"""

import torch.nn as nn

from .config import FLAGS
from .layers import NodeEmbedding, Loss
from .layers_aggregation import NodeAggregation, NodeAggregationPairs, GMNAggregatorPairs
from .layers_gmn import GMNPropagator
from .layers_link_pred import LinkPred
from .layers_load_interaction_graph import LoadInteractionGraph
from .layers_meta import MetaLayerWrapper

__all__ = ["create_layers", "layer_ctors", "parse_layer_spec"]


def parse_layer_spec(spec):
    head, _, tail = spec.partition(':')
    info = {}
    if tail:
        assert ':' not in tail
        for item in tail.split(','):
            key, _, val = item.partition('=')
            info[key] = val
    return head, info


def create_layers(model, pattern, num_layers):
    layers = nn.ModuleList()
    for i in range(1, num_layers + 1):           # 1-indexed flags
        name, info = parse_layer_spec(vars(FLAGS)['{}_{}'.format(pattern, i)])
        if name not in layer_ctors:
            raise ValueError('Unknown layer {}'.format(name))
        layers.append(layer_ctors[name](info, model, i, layers, num_layers))
    return layers


def _check_spec(allowed_nums, lf, ln):
    if len(lf) not in allowed_nums:
        raise ValueError('{} layer must have {} specs NOT {} {}'.format(ln, allowed_nums, len(lf), lf))


def _parse_as_bool(b):
    if b == 'True':
        return True
    if b == 'False':
        return False
    raise RuntimeError('Unknown bool string {}'.format(b))


def _opt_int(lf, key):
    v = lf.get(key)
    return int(v) if v else None


def _input_dim_and_level(lf, cls, layers, layer_id, model):
    higher = lf.get('higher_level')
    higher = _parse_as_bool(higher) if higher else False
    in_dim = lf.get('input_dim')
    if in_dim is not None:
        return int(in_dim), higher
    if any(type(l) is cls for l in layers):
        raise RuntimeError('The input dim for layer {} must be specified'.format(layer_id))
    return (model.interaction_num_node_feat if higher else model.num_node_feat), higher


def _node_embedding(lf, model, layer_id, layers, *unused):
    _check_spec([4, 5, 6, 7], lf, 'NodeEmbedding')
    in_dim, higher = _input_dim_and_level(lf, NodeEmbedding, layers, layer_id, model)
    return NodeEmbedding(type=lf['type'], in_dim=in_dim, out_dim=int(lf['output_dim']),
                         act=lf['act'], bn=_parse_as_bool(lf['bn']),
                         normalize=_parse_as_bool(lf['normalize']), higher_level=higher)


def _meta_layer(lf, model, layer_id, layers, *unused):
    _check_spec([5, 6], lf, 'MetaLayerWrapper')
    in_dim, higher = _input_dim_and_level(lf, MetaLayerWrapper, layers, layer_id, model)
    return MetaLayerWrapper(input_dim=in_dim, edge_dim=model.num_hyper_edge_feat,
                            output_dim=int(lf['output_dim']), edge_model=lf['edge_model'],
                            node_model=lf['node_model'], higher_level=higher,
                            num_edge_types=model.num_hyper_edge_feat, act=lf['act'])


def _load_interaction(lf, *unused):
    _check_spec([0], lf, 'LoadInteractionGraph')
    return LoadInteractionGraph()


def _node_aggregation(lf, model, layer_id, layers, num_layers, *unused):
    _check_spec([1, 3, 4], lf, "NodeAggregation")
    multi = lf.get('concat_multi_scale')
    return NodeAggregation(
        style=lf["style"],
        is_last_layer=(layer_id + 2) == num_layers,     # followed by link predictor + loss
        concat_multi_scale=_parse_as_bool(multi) if multi is not None else False,
        in_dim=_opt_int(lf, 'in_dim'), out_dim=_opt_int(lf, 'out_dim'),
        num_mlp_layers=_opt_int(lf, 'num_mlp_layers'))


def _node_aggregation_pairs(lf, *unused):
    _check_spec([1], lf, "NodeAggregationPairs")
    return NodeAggregationPairs(style=lf["style"])


def _link_pred(lf, model, *unused):
    _check_spec([3, 4, 5], lf, "LinkPred")
    multi = lf['multi_label_pred']
    return LinkPred(type=lf["type"], mlp_dim=_opt_int(lf, 'mlp_dim'),
                    weight_dim=_opt_int(lf, 'weight_dim'),
                    batch_unique_graphs=_parse_as_bool(lf["batch_unique_graphs"]),
                    multi_label_pred=_parse_as_bool(multi) if multi else False,
                    num_labels=model.num_labels + 1)


def _loss(lf, *unused):
    _check_spec([1], lf, 'Loss')
    return Loss(type=lf['type'])


def _gmn_propagator(lf, model, layer_id, layers, *unused):
    _check_spec([3, 4, 5], lf, 'GMNPropagator')
    in_dim = lf.get('input_dim')
    if not in_dim:
        if any(type(l) is GMNPropagator for l in layers):
            raise RuntimeError('The input dim for layer {} must be specified'.format(layer_id))
        in_dim = model.num_node_feat
    return GMNPropagator(input_dim=int(in_dim), output_dim=int(lf['output_dim']),
                         f_node=lf.get('f_node') or 'MLP')


def _gmn_aggregator_pairs(lf, *unused):
    _check_spec([2], lf, 'GMNAggregatorPairs')
    # the reference omits the required ``bn`` here (latent bug); bn=True is
    # what NodeAggregationPairs passes for the same module.
    return GMNAggregatorPairs(input_dim=int(lf['input_dim']), output_dim=int(lf['output_dim']), bn=True)


layer_ctors = {
    'NodeEmbedding': _node_embedding,
    'NodeAggregation': _node_aggregation,
    'NodeAggregationPairs': _node_aggregation_pairs,
    'Loss': _loss,
    'GMNPropagator': _gmn_propagator,
    'GMNAggregatorPairs': _gmn_aggregator_pairs,
    'LinkPredictor': _link_pred,
    'MetaLayer': _meta_layer,
    'LoadInteractionLayer': _load_interaction,
}
