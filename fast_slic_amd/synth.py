"""Deterministic synthetic RGB frames (integer-only, libm-free, RNG-stream-free).

Generator specified in SURVEY.md section 8(d).  Used by tests, bench.py and the golden
fixture generator so that the container, the GPU box and any later round see the same
bytes.  Variants:
  A  noise_bits=5, blocks=True   (primary; fragments heavily, CCA top-K path taken)
  B  noise_bits=0, blocks=False  (clean gradients)
  C  uniform noise               (CCA worst case)
  D  all zeros                   (maximal distance ties, reference test/test_slic.py:42)
"""
import numpy as np


def _hash(H, W, c, seed):
    y = np.arange(H, dtype=np.uint64)[:, None]
    x = np.arange(W, dtype=np.uint64)[None, :]
    h = (x * np.uint64(73856093)) ^ (y * np.uint64(19349663)) ^ np.uint64(
        (c + 1) * 83492791 + seed * 2654435761)
    h = ((h ^ (h >> np.uint64(13))) * np.uint64(0x5bd1e995)) & np.uint64(0xFFFFFFFF)
    return h ^ (h >> np.uint64(15))


def synth(H, W, seed=0, noise_bits=5, blocks=True):
    yi = np.arange(H, dtype=np.int64)[:, None]
    xi = np.arange(W, dtype=np.int64)[None, :]
    out = np.empty((H, W, 3), dtype=np.uint8)
    for c in range(3):
        h = _hash(H, W, c, seed)
        if noise_bits > 0:
            noise = (h & np.uint64((1 << noise_bits) - 1)).astype(np.int64) - (1 << (noise_bits - 1))
        else:
            noise = np.zeros((H, W), dtype=np.int64)
        tri = np.abs(((xi * (3 + c) + yi * (2 * c + 1)) % 512) - 256) * 200 // 256
        if blocks:
            blk = ((yi // 48 + (xi // 64) * 7 + 3 * c) * 37) % 64
        else:
            blk = 0
        out[:, :, c] = np.clip(tri + blk + noise, 0, 255).astype(np.uint8)
    return out


def synth_noise(H, W, seed=0):
    """Variant C: uniform byte noise."""
    out = np.empty((H, W, 3), dtype=np.uint8)
    for c in range(3):
        out[:, :, c] = (_hash(H, W, c, seed) & np.uint64(255)).astype(np.uint8)
    return out


def variant(name, H, W, seed=0):
    if name == "A":
        return synth(H, W, seed, 5, True)
    if name == "B":
        return synth(H, W, seed, 0, False)
    if name == "C":
        return synth_noise(H, W, seed)
    if name == "D":
        return np.zeros((H, W, 3), np.uint8)
    raise ValueError(name)
