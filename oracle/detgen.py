"""Deterministic, platform-independent tensor generator (TEST INFRASTRUCTURE).

Golden fixtures under ``tests/golden/`` store only *outputs* of the reference;
inputs and weights are regenerated bit-exactly from ``(seed, name, shape)`` with
a splitmix64 integer hash, so they do not depend on any library RNG stream.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed: int, name: str, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    """float32 array, uniform in [lo, hi), a pure function of (seed, name, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    tag = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    base = (np.uint64(seed) << np.uint64(32)) ^ tag
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + base
        bits = _splitmix64(idx) >> np.uint64(40)          # 24 random bits
    u = bits.astype(np.float64) / float(1 << 24)           # [0,1) exactly representable
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def passt_state_dict(cfg: dict, seed: int) -> dict:
    """Randomised (test-sensitive) parameters with the reference's state_dict schema
    (reference models/passt.py:429-467).  Every tensor -- including LN
    gains/biases, linear biases and positional embeddings -- is non-trivial so that a
    wrong kernel cannot hide behind an identity/zero parameter."""
    D = cfg["embed_dim"]
    depth = cfg["depth"]
    C = cfg["num_classes"]
    Fg, Tg = cfg["grid"]
    P = cfg.get("patch", 16)
    hid = 4 * D
    sd = {}

    def put(name, shape, scale, shift=0.0):
        sd[name] = uniform(seed, name, shape, -scale, scale) + np.float32(shift)

    put("cls_token", (1, 1, D), 0.5)
    put("dist_token", (1, 1, D), 0.5)
    put("new_pos_embed", (1, 2, D), 0.5)
    put("freq_new_pos_embed", (1, D, Fg, 1), 0.5)
    put("time_new_pos_embed", (1, D, 1, Tg), 0.5)
    put("patch_embed.proj.weight", (D, 1, P, P), 1.0 / 16)
    put("patch_embed.proj.bias", (D,), 0.2)
    for i in range(depth):
        p = f"blocks.{i}."
        put(p + "norm1.weight", (D,), 0.3, 1.0)
        put(p + "norm1.bias", (D,), 0.2)
        put(p + "attn.qkv.weight", (3 * D, D), 1.5 / np.sqrt(D))
        put(p + "attn.qkv.bias", (3 * D,), 0.2)
        put(p + "attn.proj.weight", (D, D), 1.0 / np.sqrt(D))
        put(p + "attn.proj.bias", (D,), 0.2)
        put(p + "norm2.weight", (D,), 0.3, 1.0)
        put(p + "norm2.bias", (D,), 0.2)
        put(p + "mlp.fc1.weight", (hid, D), 1.5 / np.sqrt(D))
        put(p + "mlp.fc1.bias", (hid,), 0.2)
        put(p + "mlp.fc2.weight", (D, hid), 1.0 / np.sqrt(hid))
        put(p + "mlp.fc2.bias", (D,), 0.2)
    put("norm.weight", (D,), 0.3, 1.0)
    put("norm.bias", (D,), 0.2)
    put("head.0.weight", (D,), 0.3, 1.0)
    put("head.0.bias", (D,), 0.2)
    put("head.1.weight", (C, D), 1.0 / np.sqrt(D))
    put("head.1.bias", (C,), 0.2)
    put("head_dist.weight", (C, D), 1.0 / np.sqrt(D))
    put("head_dist.bias", (C,), 0.2)
    return sd
