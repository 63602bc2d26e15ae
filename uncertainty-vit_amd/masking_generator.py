"""Block-wise mask generator of BEiT / data2vec pre-training: the contract of the reference's
masking_generator.py:29-92 ((H, W) 0/1 mask with AT MOST `num_masking_patches` ones, built from random
rectangles of `min_num_patches`..`max_num_patches` cells with log-uniform aspect ratio in [0.3, 1/0.3]).

Differences from the reference, all deliberate:
  * the generator owns a seeded `random.Random` (the reference draws from the unseeded global `random`,
    run_cyclical.py:318 is commented out);
  * the mask dtype is int64 (`np.int` no longer exists);
  * the grid is held as ONE Python integer (bit r*W + c = cell (r, c)): a rectangle is a precomputed bit pattern, the
    overlap count is `int.bit_count`, painting is an OR -- no per-cell loop and no array slicing inside the rejection
    loop -- and `batch(n)` draws n masks and unpacks them into one (n, H, W) array (the loader-side call per step).
Given the same random stream the masks are bit-identical to the reference's (tests/golden/masks.npz).
"""
import math
import random

import numpy as np


class MaskingGenerator:
    def __init__(self, input_size, num_masking_patches, min_num_patches=4, max_num_patches=None, min_aspect=0.3,
                 max_aspect=None, seed=None):
        self.height, self.width = input_size if isinstance(input_size, tuple) else (input_size, input_size)
        self.num_patches = self.height * self.width
        self.num_masking_patches = num_masking_patches
        self.min_num_patches = min_num_patches
        self.max_num_patches = max_num_patches if max_num_patches is not None else num_masking_patches
        hi = max_aspect or 1 / min_aspect
        self.log_aspect_ratio = (math.log(min_aspect), math.log(hi))
        self.rng = random.Random(seed)
        self._row = [(1 << w) - 1 for w in range(self.width + 1)]           # w adjacent cells of one grid row
        self._rect = {}                                                     # (h, w) -> that rectangle at the origin

    def __repr__(self):
        lo, hi = (math.exp(v) for v in self.log_aspect_ratio)
        return (f"MaskingGenerator(grid={self.height}x{self.width}, blocks of {self.min_num_patches}..{self.max_num_patches} "
                f"cells, at most {self.num_masking_patches} masked, aspect {lo:.2f}..{hi:.2f})")

    def get_shape(self):
        return self.height, self.width

    def _rect_bits(self, h, w):
        r = self._rect.get((h, w))
        if r is None:
            r = 0
            for i in range(h):
                r |= self._row[w] << (i * self.width)
            self._rect[(h, w)] = r
        return r

    def _add_block(self, grid, budget):
        """Up to 10 rectangle draws (same order of random calls as masking_generator.py:54-66); paints the first one that
        adds between 1 and `budget` new cells.  Returns (grid, cells added)."""
        rng, H, W = self.rng, self.height, self.width
        for _ in range(10):
            area = rng.uniform(self.min_num_patches, budget)
            aspect = math.exp(rng.uniform(*self.log_aspect_ratio))
            h, w = int(round(math.sqrt(area * aspect))), int(round(math.sqrt(area / aspect)))
            if w < W and h < H:
                top, left = rng.randint(0, H - h), rng.randint(0, W - w)
                rect = self._rect_bits(h, w) << (top * W + left)
                new = h * w - (grid & rect).bit_count()
                if 0 < new <= budget:
                    return grid | rect, new
        return grid, 0

    def _draw(self):
        grid, count = 0, 0
        while count < self.num_masking_patches:
            grid, added = self._add_block(grid, min(self.num_masking_patches - count, self.max_num_patches))
            if added == 0:
                break
            count += added
        return grid

    def _unpack(self, grids):
        nbytes = (self.num_patches + 7) // 8
        raw = np.frombuffer(b"".join(g.to_bytes(nbytes, "little") for g in grids), dtype=np.uint8).reshape(len(grids), nbytes)
        bits = np.unpackbits(raw, axis=1, bitorder="little")[:, :self.num_patches]
        return bits.reshape(len(grids), self.height, self.width).astype(np.int64)

    def __call__(self):
        return self._unpack([self._draw()])[0]

    def batch(self, n):
        """n masks from the generator's stream as one (n, H, W) int64 array: identical to n successive calls."""
        return self._unpack([self._draw() for _ in range(n)])
