"""Block-wise mask generator of BEiT / data2vec pre-training: the contract of the reference's
masking_generator.py:29-92 ((H, W) 0/1 mask with AT MOST `num_masking_patches` ones, built from random
rectangles of `min_num_patches`..`max_num_patches` cells with log-uniform aspect ratio in [0.3, 1/0.3]).

Differences from the reference, both deliberate: the generator owns a seeded `random.Random` (the reference
draws from the unseeded global `random`, run_cyclical.py:318 is commented out), and the mask dtype is int64
(`np.int` no longer exists).  Given the same random stream the masks are identical (tests/golden/masks.npz).
"""
import math
import random

import numpy as np


class MaskingGenerator:
    def __init__(self, input_size, num_masking_patches, min_num_patches=4, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None, seed=None):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 2
        self.height, self.width = input_size
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = num_masking_patches if max_num_patches is None else max_num_patches
        max_aspect = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(max_aspect))
        self.rng = random.Random(seed)

    def __repr__(self):
        return "Generator(%d, %d -> [%d ~ %d], max = %d, %.3f ~ %.3f)" % (
            self.height, self.width, self.min_num_patches, self.max_num_patches, self.num_masking_patches,
            self.log_aspect_ratio[0], self.log_aspect_ratio[1])

    def get_shape(self):
        return self.height, self.width

    def _add_block(self, mask, budget):
        """Try up to 10 rectangles; paint the first one that adds between 1 and `budget` new cells."""
        for _ in range(10):
            area = self.rng.uniform(self.min_num_patches, budget)
            aspect = math.exp(self.rng.uniform(*self.log_aspect_ratio))
            h, w = int(round(math.sqrt(area * aspect))), int(round(math.sqrt(area / aspect)))
            if w < self.width and h < self.height:
                top, left = self.rng.randint(0, self.height - h), self.rng.randint(0, self.width - w)
                window = mask[top:top + h, left:left + w]
                new = h * w - int(window.sum())
                if 0 < new <= budget:
                    window[...] = 1
                    return new
        return 0

    def __call__(self):
        mask = np.zeros(self.get_shape(), dtype=np.int64)
        count = 0
        while count < self.num_masking_patches:
            added = self._add_block(mask, min(self.num_masking_patches - count, self.max_num_patches))
            if added == 0:
                break
            count += added
        return mask
