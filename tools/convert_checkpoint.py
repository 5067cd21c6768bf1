#!/usr/bin/env python3
"""One-off converter: the reference's whole-module pickles -> plain `state_dict` files.

The reference saves and loads *entire module objects* (utils/util.py:121-133 `torch.save(model, path, pickle_protocol=4)`,
:135-159 `torch.load(path)`; train.py:377-388 saves the best Swin and multimodal model, train.py:428-432 loads
`pretrained_model/<file>` for --doEval).  During training both are wrapped by LightningLite (`self.setup(...)`,
train.py:331), so the published files un-pickle to `pytorch_lightning.lite.wrappers._LiteModule` objects around
`src.models.SwinForAffwildClassification` / `MultiModalTransformerForClassification`: un-pickling needs
pytorch_lightning 1.8.3 *and* the reference's own source tree on sys.path.  Neither ships with facialmmt_amd.

Run this ONCE, in the reference's environment (the one its requirements.txt describes):

    python tools/convert_checkpoint.py --reference /path/to/FacialMMT \\
        --in  /path/to/FacialMMT/pretrained_model/best_swin_XX.pt --out best_swin.state.pt

and then, anywhere (no reference, no lightning):

    from facialmmt_amd import checkpoint, models
    m = models.SwinForAffwildClassification(args)           # or MultiModalTransformerForClassification
    checkpoint.load_state(m, "best_swin.state.pt")          # strict: every key and shape must match

What it does: puts the reference tree on sys.path, `torch.load(..., weights_only=False)` (this executes the pickle:
only do it with files you trust -- exactly what the reference itself does), unwraps `_LiteModule` /
`DataParallel` (`_forward_module.` / `module.` prefixes) with `checkpoint.extract_state_dict`, and writes
`{'state_dict': {name: cpu tensor}, 'source': ..., 'class': ...}` with tensors and builtins only, so the result loads
with `torch.load(weights_only=True)`.  A FaceX-Zoo backbone file (`{'state_dict': {'backbone.*': ...}}`,
train.py:316-331) is already plain and needs no conversion: use `checkpoint.load_pretrained_backbone`.
"""
from __future__ import annotations

import argparse
import os
import sys


def convert(src_path: str, out_path: str, reference_root: str | None = None, expect_class: str | None = None) -> dict:
    """Returns a small report {class, n_tensors, n_params, stripped_prefixes}.  Raises if the file holds no module /
    mapping, or if `expect_class` is given and the innermost module's class name differs."""
    import torch

    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if here not in sys.path:
        sys.path.insert(0, here)
    from facialmmt_amd import checkpoint

    if reference_root:
        reference_root = os.path.abspath(reference_root)
        if not os.path.isdir(reference_root):
            raise FileNotFoundError(f"--reference {reference_root}: not a directory")
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)             # the pickle names src.models.*, modules.* of the reference
    obj = torch.load(src_path, map_location="cpu", weights_only=False)
    inner = obj
    seen = []
    # LightningLite's _LiteModule keeps the user's module in `_forward_module` (older versions: `module`); DataParallel
    # and DistributedDataParallel in `module`
    while isinstance(inner, torch.nn.Module):
        nxt = getattr(inner, "_forward_module", None)
        if not isinstance(nxt, torch.nn.Module):
            nxt = inner.module if type(inner).__name__ in ("DataParallel", "DistributedDataParallel", "_LiteModule") and \
                isinstance(getattr(inner, "module", None), torch.nn.Module) else None
        if nxt is None:
            break
        seen.append(type(inner).__name__)
        inner = nxt
    cls = type(inner).__name__
    if expect_class and cls != expect_class:
        raise TypeError(f"{src_path} holds a {cls}, expected {expect_class}")
    sd = checkpoint.extract_state_dict(inner if isinstance(inner, torch.nn.Module) else obj)
    payload = {"state_dict": sd, "source": os.path.basename(src_path), "class": cls, "wrappers": list(seen)}
    tmp = f"{out_path}.tmp.{os.getpid()}"
    torch.save(payload, tmp)
    os.replace(tmp, out_path)
    torch.load(out_path, map_location="cpu", weights_only=True)      # prove the result needs no code to load
    return {"class": cls, "n_tensors": len(sd), "n_params": int(sum(v.numel() for v in sd.values())), "wrappers": seen}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--in", dest="src", required=True, help="whole-module pickle written by the reference (utils/util.py:121-133)")
    ap.add_argument("--out", required=True, help="plain state_dict file to write")
    ap.add_argument("--reference", default=None, help="root of the reference source tree (needed to un-pickle its classes)")
    ap.add_argument("--expect-class", default=None, help="fail unless the innermost module has this class name")
    a = ap.parse_args(argv)
    rep = convert(a.src, a.out, a.reference, a.expect_class)
    print(f"{a.out}: {rep['class']} ({'/'.join(rep['wrappers']) or 'unwrapped'}), {rep['n_tensors']} tensors, {rep['n_params'] / 1e6:.2f} M values")
    return 0


if __name__ == "__main__":
    sys.exit(main())
