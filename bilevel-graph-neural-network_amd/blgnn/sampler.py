"""``RandomSampler`` -- the reference's default pair-batch sampler (src/sampler.py:110-131):
a shuffled pass over the training pairs in chunks of ``batch_size``, restarted when
exhausted; returns ``(batch_gids [B,2], sampled_gids (unique), None)``.  The structural
samplers (NeighborSampler / EverythingSampler) are outside the hot path."""

import numpy as np

__all__ = ["RandomSampler"]


class RandomSampler(object):
    def __init__(self, data, batch_size, sample_induced=False, seed=None):
        if sample_induced:
            raise NotImplementedError("sample_induced is off in every shipped config (src/config.py:153)")
        self.pairs = np.asarray(data.dataset.train_pairs, dtype=np.int64).reshape(-1, 2)
        self.batch_size = int(batch_size)
        self.rng = np.random.default_rng(seed)
        self._order = None
        self._pos = 0

    def _reshuffle(self):
        self._order = self.rng.permutation(len(self.pairs))
        self._pos = 0

    def sample_next_training_batch(self):
        if self._order is None or self._pos >= len(self._order):
            self._reshuffle()
        idx = self._order[self._pos:self._pos + self.batch_size]
        self._pos += self.batch_size
        batch_gids = self.pairs[idx]
        return batch_gids, np.unique(batch_gids), None
