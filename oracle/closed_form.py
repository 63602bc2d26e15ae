"""ORACLE helper — test infrastructure, NOT product code.

Closed-form pseudo-random tensors that the fixture generator (tools/gen_golden.py, which runs
the reference) and the tests evaluate identically, so ViT-shaped weights and inputs never need
to be committed: value[i] comes from an integer hash of i keyed by the tensor's name.
"""
import hashlib

import numpy as np
import torch


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def closed_form(name: str, shape, scale: float = 0.02) -> torch.Tensor:
    """Uniform on [-scale*sqrt(3), scale*sqrt(3)] (std = scale) from an integer hash of the element
    index keyed by the tensor name: bit-reproducible on any host (integer arithmetic only)."""
    key = np.uint32(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        h = _mix32(_mix32(np.arange(n, dtype=np.uint32) ^ key) + np.uint32(0x9E3779B9))
    u = (h.astype(np.float64) + 0.5) / 4294967296.0
    v = (scale * np.sqrt(3.0) * (2.0 * u - 1.0)).astype(np.float32)
    return torch.from_numpy(v).reshape(tuple(shape))


def closed_form_state(shapes: dict, table_scale: float = 0.05, gamma: float = 0.1) -> dict:
    """State dict for the base model: name -> tensor, given name -> shape (float entries only)."""
    out = {}
    for k, shape in shapes.items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            out[k] = 1.0 + closed_form(k, shape, 0.1)
        elif "gamma" in k:
            out[k] = gamma * (1.0 + closed_form(k, shape, 0.5))
        elif k.endswith("relative_position_bias_table"):
            out[k] = closed_form(k, shape, table_scale)
        elif k.endswith("bias") or k.endswith("q_bias") or k.endswith("v_bias"):
            out[k] = closed_form(k, shape, 0.01)
        else:
            out[k] = closed_form(k, shape, 0.03)
    return out


def closed_form_images(tag: str, B: int, img: int, scale: float = 1.0) -> torch.Tensor:
    return closed_form("images/" + tag, (B, 3, img, img), scale)


def exact_masks(B: int, n_patches: int, n_mask: int, seed: int) -> torch.Tensor:
    """(B, g, g) int64 with exactly n_mask ones per image (seeded randperm)."""
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(B, n_patches, dtype=torch.int64)
    for b in range(B):
        m[b, torch.randperm(n_patches, generator=g)[:n_mask]] = 1
    side = int(round(n_patches ** 0.5))
    return m.reshape(B, side, side)


def checksum(t: torch.Tensor, n_samples: int = 64):
    """(sum, abs-sum) in float64 plus n_samples evenly spaced elements."""
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n_samples, f.numel())).long()
    return np.array([f.sum().item(), f.abs().sum().item()]), f[idx].float().numpy()
