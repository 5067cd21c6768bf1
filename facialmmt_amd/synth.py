"""Torch-independent deterministic tensor generator (integer hash -> float).

Every synthetic weight and input used by the golden fixtures, the parity tests,
``__graft_entry__.smoke()`` and ``bench.py`` comes from here, so that the GPU box
(which never sees ``/root/reference``) regenerates bit-identical fp32 tensors from
nothing but a name, a shape and a seed.

The generator is splitmix64 keyed by FNV-1a(name) ^ f(seed); element ``i`` of a
tensor is a pure function of (name, seed, i), so shapes may be regenerated lazily
and sliced freely.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _fnv1a64(name: str) -> np.uint64:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return np.uint64(h)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + _GOLD) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def _key(name: str, seed: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return _splitmix64(np.array([_fnv1a64(name) ^ (np.uint64(seed) * _GOLD)], dtype=np.uint64))[0]


def unit(name: str, shape, seed: int = 0) -> np.ndarray:
    """float32 array, i.i.d.-looking uniform in [0, 1) with 24-bit resolution."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * _GOLD + _key(name, seed)
    bits = _splitmix64(idx) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(tuple(shape))


def uniform(name: str, shape, seed: int = 0, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    u = unit(name, shape, seed)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def randint(name: str, shape, lo: int, hi: int, seed: int = 0) -> np.ndarray:
    """int64 array uniform in [lo, hi)."""
    u = unit(name, shape, seed).astype(np.float64)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64)


# --------------------------------------------------------------------------------------
# state_dict filling
# --------------------------------------------------------------------------------------

def _rule(key: str, shape):
    """(lo, hi) used for a state_dict entry, chosen so that activations stay O(1)
    through 12 Swin blocks and every learned term (biases, LayerNorm affine, relative
    position table, BatchNorm running stats) is visibly non-trivial in the parity tests."""
    leaf = key.rsplit(".", 1)[-1]
    if "relative_position_bias_table" in key:
        return (-0.5, 0.5)
    if leaf == "running_var":
        return (0.5, 1.5)
    if leaf == "running_mean":
        return (-0.2, 0.2)
    if leaf == "query_vector":
        return (-1.0, 1.0)
    is_norm = ("norm" in key.lower()) or key.startswith("output_layer.0.") or key.startswith("output_layer.3.") \
        or ".output_layer.0." in key or ".output_layer.3." in key
    if is_norm and leaf == "weight" and len(shape) == 1:
        return (0.8, 1.2)
    if leaf in ("bias", "in_proj_bias"):
        return (-0.1, 0.1)
    if leaf in ("weight", "in_proj_weight") and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = float(np.sqrt(3.0 / fan_in))  # unit-gain uniform
        return (-a, a)
    return (-0.1, 0.1)


STRUCTURAL = ("relative_position_index", "attn_mask", "num_batches_tracked", "version", "_float_tensor",
              "position_ids", "token_type_ids")


def fill_state_dict(module, seed: int = 0, prefix: str = ""):
    """Overwrite every floating parameter/buffer of ``module`` in place with hash-derived values.

    Structural buffers (index tables, shift masks, counters) are left as constructed.
    Returns the module."""
    import torch

    sd = module.state_dict()
    with torch.no_grad():
        for k, t in sd.items():
            if any(s in k for s in STRUCTURAL) or not t.is_floating_point():
                continue
            lo, hi = _rule(k, tuple(t.shape))
            arr = uniform(prefix + k, tuple(t.shape), seed, lo, hi)
            t.copy_(torch.from_numpy(arr).to(t.dtype))
    return module


def state_dict_from_keys(keys, seed: int = 0, prefix: str = "", device=None, dtype=None):
    """Rebuild the float entries of a state_dict from a [[key, shape, dtype], ...] list exactly as
    fill_state_dict would have filled a module with those keys (structural buffers are skipped)."""
    import torch

    out = {}
    for k, shape, dt in keys:
        if any(s in k for s in STRUCTURAL) or "float" not in dt:
            continue
        lo, hi = _rule(k, tuple(shape))
        t = torch.from_numpy(uniform(prefix + k, tuple(shape), seed, lo, hi))
        out[k] = t.to(device=device, dtype=dtype) if (device is not None or dtype is not None) else t
    return out


def tensor(name: str, shape, seed: int = 0, lo: float = -1.0, hi: float = 1.0, device=None, dtype=None):
    import torch

    t = torch.from_numpy(uniform(name, shape, seed, lo, hi))
    if device is not None or dtype is not None:
        t = t.to(device=device, dtype=dtype)
    return t


def make_standin_plm(vocab: int = 1000, hidden: int = 1024, seed: int = 7):
    """Deterministic stand-in for the HuggingFace text encoder used by the parity fixtures:
    an embedding table -> (B, T, hidden) zeroed where the attention mask is 0.  It exposes
    `.config.hidden_size` and returns a tuple like the HF models.  Not a model of RoBERTa: it only makes
    the *surrounding* arithmetic (slicing, fusion, pooling) comparable between reference and build."""
    import torch
    import torch.nn as nn
    from types import SimpleNamespace

    class StandInPLM(nn.Module):
        def __init__(self):
            super().__init__()
            self.config = SimpleNamespace(hidden_size=hidden)
            self.emb = nn.Embedding(vocab, hidden)
            with torch.no_grad():
                self.emb.weight.copy_(torch.from_numpy(uniform("standin_plm.emb", (vocab, hidden), seed, -1.0, 1.0)))

        def forward(self, input_ids, attention_mask=None, **kw):
            h = self.emb(input_ids)
            if attention_mask is not None:
                h = h * attention_mask.unsqueeze(-1).to(h.dtype)
            return (h,)

    return StandInPLM()
