"""Checkpoint I/O for the hot-path modules: plain `state_dict` files in, plain
`state_dict` files out.  Host-side only; nothing here touches the GPU or the HIP library.

What the reference does and what replaces it:

* train.py:316-331 initialises `SwinForAffwildClassification` from FaceX-Zoo's `Swin_tiny_Ms-Celeb-1M.pt`:
  a dict with a `state_dict` entry whose backbone weights are stored as `backbone.<name>`; the model's own
  keys are `swin.<name>` for the backbone and bare names for the head; `classifier.*` is never taken from the
  file.  -> `load_pretrained_backbone`.
* utils/util.py:121-159 pickles whole `nn.Module` objects (`torch.save(model)`), wrapped by Lightning Lite's
  `_LiteModule` during training, so the published checkpoints need the reference's classes and
  pytorch_lightning 1.8.3 to un-pickle.  Neither is shipped with this package; once such an object has been
  un-pickled in the reference's own environment, `extract_state_dict` turns it (or its `state_dict()`) into
  the plain mapping our same-named modules load with `load_state_dict(strict=True)`: the parameter names and
  shapes are identical by construction (tests/test_host_cpu.py::test_state_dict_keys_match_reference).
  -> `extract_state_dict`, `save_state`, `load_state`.

The reference's loop tests `k in pretrained_dict` with the *model's* key before reading
`pretrained_dict['backbone.' + name]` (train.py:320-329), which only matches files that hold both spellings;
here a tensor is taken whenever `backbone.<name>` is present, and what was and was not found is returned so the
caller can see it rather than silently training from random weights."""
from __future__ import annotations

import collections
import os
from typing import Dict, Iterable, Mapping, NamedTuple

import torch

# wrapper prefixes a pickled training-time module accumulates: nn.DataParallel / DDP ("module."), Lightning
# Lite's _LiteModule ("_forward_module." / "module.") -- main.py:152-160 runs under strategy='dp'
_WRAPPER_PREFIXES = ("_forward_module.", "module.")
_SKIP_FROM_PRETRAINED = ("classifier.weight", "classifier.bias")       # train.py:322-325


class LoadReport(NamedTuple):
    loaded: list          # model keys that were filled from the file
    missing: list         # model keys the file has no tensor for (kept at their initial values)
    unused: list          # file keys nothing in the model asked for
    mismatched: list      # (key, shape in file, shape in model): never loaded


def _strip(key: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in _WRAPPER_PREFIXES:
            if key.startswith(p):
                key, changed = key[len(p):], True
    return key


def extract_state_dict(obj) -> "collections.OrderedDict[str, torch.Tensor]":
    """`obj`: an un-pickled reference model (possibly wrapped), its `state_dict()`, or a FaceX-Zoo style
    `{'state_dict': ...}` file content.  Returns an ordered {name: detached CPU tensor} with wrapper prefixes
    removed.  Raises on key collisions after stripping (two different wrappers in one mapping)."""
    if isinstance(obj, torch.nn.Module):
        obj = obj.state_dict()
    if isinstance(obj, Mapping) and "state_dict" in obj and isinstance(obj["state_dict"], Mapping):
        obj = obj["state_dict"]
    if not isinstance(obj, Mapping):
        raise TypeError(f"cannot take a state_dict from {type(obj).__name__}")
    out = collections.OrderedDict()
    for k, v in obj.items():
        if not isinstance(v, torch.Tensor):
            raise TypeError(f"entry {k!r} is {type(v).__name__}, not a tensor")
        nk = _strip(k)
        if nk in out:
            raise KeyError(f"{k!r} collides with another entry after removing wrapper prefixes")
        out[nk] = v.detach().to("cpu")
    return out


def load_pretrained_backbone(model, checkpoint, skip: Iterable[str] = _SKIP_FROM_PRETRAINED) -> LoadReport:
    """Fill `model` (SwinForAffwildClassification, or a bare backbone) from a FaceX-Zoo checkpoint
    (train.py:316-331).  `checkpoint`: a path, the loaded dict, or its `state_dict`.

    model key `swin.<name>` / `<name>`  <-  file key `backbone.<name>`; keys in `skip` are left alone.  A
    shape mismatch is reported and not loaded (the reference would raise inside load_state_dict; a classifier
    head of a different width is the one legitimate case and it is skipped by name already)."""
    if isinstance(checkpoint, (str, os.PathLike)):
        checkpoint = torch.load(checkpoint, map_location="cpu", weights_only=True)
    src = extract_state_dict(checkpoint)
    own = model.state_dict()
    skip = set(skip)
    loaded, missing, mismatched, used = [], [], [], set()
    new = {}
    for k, cur in own.items():
        if k in skip:
            continue
        name = k[5:] if k.startswith("swin.") else k
        fk = "backbone." + name
        if fk not in src:
            missing.append(k)
            continue
        used.add(fk)
        t = src[fk]
        if tuple(t.shape) != tuple(cur.shape):
            mismatched.append((k, tuple(t.shape), tuple(cur.shape)))
            continue
        new[k] = t.to(dtype=cur.dtype)
        loaded.append(k)
    merged = dict(own)
    merged.update(new)
    model.load_state_dict(merged, strict=True)
    unused = [k for k in src if k not in used]
    return LoadReport(loaded, missing, unused, mismatched)


def save_state(model, path: str, extra: Dict[str, object] | None = None) -> None:
    """Write `{'state_dict': plain CPU state_dict, **extra}` -- tensors and builtins only, loadable with
    `torch.load(weights_only=True)` and by the reference's own classes via load_state_dict (replaces the
    whole-module pickles of utils/util.py:121-133,143-147)."""
    payload = {"state_dict": extract_state_dict(model)}
    for k, v in (extra or {}).items():
        if k == "state_dict":
            raise KeyError("'state_dict' is reserved")
        payload[k] = v
    tmp = f"{path}.tmp.{os.getpid()}"
    torch.save(payload, tmp)
    os.replace(tmp, path)                                  # a best-on-validation file is never half-written


def load_state(model, source, strict: bool = True) -> LoadReport:
    """Load a file written by `save_state`, a reference `state_dict`, or an un-pickled reference module into the
    same-named module of this package (replaces utils/util.py:135-141,149-159).  strict=True raises on any
    missing / unexpected / mis-shaped key, like load_state_dict."""
    if isinstance(source, (str, os.PathLike)):
        source = torch.load(source, map_location="cpu", weights_only=True)
    src = extract_state_dict(source)
    own = model.state_dict()
    missing = [k for k in own if k not in src]
    unused = [k for k in src if k not in own]
    mismatched = [(k, tuple(src[k].shape), tuple(own[k].shape)) for k in own if k in src and tuple(src[k].shape) != tuple(own[k].shape)]
    if strict and (missing or unused or mismatched):
        raise RuntimeError(f"state_dict does not fit {type(model).__name__}: missing {missing[:5]} "
                           f"unexpected {unused[:5]} mis-shaped {mismatched[:5]}")
    bad = {k for k, _, _ in mismatched}
    new = dict(own)
    loaded = []
    for k in own:
        if k in src and k not in bad:
            new[k] = src[k].to(dtype=own[k].dtype)
            loaded.append(k)
    model.load_state_dict(new, strict=True)
    return LoadReport(loaded, missing, unused, mismatched)
