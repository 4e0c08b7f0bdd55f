"""The callers of the hot path: what src/train.py does around ``Model.forward``.

``get_initial_embd`` mirrors train.py:48-72 (sweep ALL molecules through the lower
level and write the pooled rows into the interaction graph's ``init_x``),
``model_forward`` mirrors train.py:75-108 (build the batch, pick the layer slice),
``bilevel_forward`` is the whole forward the north-star metric times: lower sweep ->
readout -> interaction-level conv -> pair scores.
"""

import numpy as np
import torch

from .batch import BatchData
from .config import FLAGS

__all__ = ["get_initial_embd", "model_forward", "bilevel_forward", "all_molecule_pairs"]


def all_molecule_pairs(gids):
    """train.py:52-59: consecutive gids paired up so that every molecule appears."""
    gids = list(gids)
    pairs = [(gids[i], gids[i + 1]) for i in range(0, len(gids) - 2, 2)]
    pairs.append((gids[len(gids) - 2], gids[len(gids) - 1]))
    return np.asarray(pairs, dtype=np.int64)


def get_initial_embd(train_data, model, chunk_pairs=None, cached_batches=None):
    """Lower-level sweep over every molecule.  ``chunk_pairs`` = pairs per chunk
    (train.py:60 uses FLAGS.batch_size; None = one chunk, which is what 288 GB of HBM
    is for).  ``cached_batches``: list reused across calls -- the molecule set is
    static, so its merged graph + CSR are built once."""
    model.train_data = train_data
    model.use_layers = 'lower_layers'
    if cached_batches is not None and len(cached_batches) > 0:
        batches = cached_batches
    else:
        allp = all_molecule_pairs(train_data.dataset.gs_map.keys())
        step = len(allp) if not chunk_pairs else int(chunk_pairs)
        batches = [BatchData(allp[i:i + step], train_data.dataset, is_train=False, ignore_pairs=True)
                   for i in range(0, len(allp), step)]
        if cached_batches is not None:
            cached_batches.extend(batches)
    # a sweep starts from values, not from the previous sweep's autograd graph: detach once here, so that inside
    # the sweep every chunk's pooled rows stay attached (layers_aggregation.py: NodeAggregation.forward)
    graph = getattr(train_data.dataset, "interaction_combo_nxgraph", None)
    if graph is not None and isinstance(getattr(graph, "init_x", None), torch.Tensor) and graph.init_x.requires_grad:
        graph.init_x = graph.init_x.detach()
    out = [model(b) for b in batches]
    return out[0] if len(out) == 1 else torch.cat(out, dim=0)


def model_forward(model, data, batch_gids, is_train=False, sampled_gids=None, rng=None,
                  lower_cache=None):
    """train.py:75-108 for an explicit pair batch."""
    bilevel = FLAGS.lower_level_layers and FLAGS.higher_level_layers
    if bilevel and "model_init" in FLAGS.init_embds:
        get_initial_embd(data, model, cached_batches=lower_cache)
        data.dataset.init_interaction_graph_embds(device=FLAGS.device)
    batch_data = BatchData(batch_gids, data.dataset, is_train=is_train, sampled_gids=sampled_gids,
                           unique_graphs=FLAGS.batch_unique_graphs, rng=rng)
    model.use_layers = 'higher_layers' if bilevel else 'all'
    return batch_data


def bilevel_forward(model, data, batch_gids, lower_cache=None, run_loss=False):
    """One complete forward; returns the pair predictions [B, 1] (or [B, L])."""
    batch_data = model_forward(model, data, batch_gids, is_train=False, lower_cache=lower_cache)
    layers = model._selected()
    model.acts = [batch_data.merge_data['merge'].x]
    for layer in (layers if run_loss else layers[:-1]):
        model.acts.append(layer(model.acts[-1], batch_data, model))
    return model.acts[-1], batch_data
