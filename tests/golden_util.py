"""Loading / comparing the committed golden fixtures (tests/golden/*.npz).

A fixture entry is either a full tensor or {strided sample, fp64 sum, fp64 abs-sum, shape}
(see oracle/gen_golden.py::pack).  `check` applies the same packing to a candidate tensor."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self):
        self.files = {n[:-4]: np.load(os.path.join(GOLD, n)) for n in os.listdir(GOLD) if n.endswith(".npz")}
        with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
            self.keys = json.load(f)

    def has(self, file, name):
        return f"{name}/shape" in self.files[file]

    def expected(self, file, name):
        z = self.files[file]
        if f"{name}/full" in z:
            return z[f"{name}/full"], None
        return z[f"{name}/sample"], int(z[f"{name}/stride"])

    def check(self, file, name, cand: torch.Tensor, atol=1e-3, rtol=1e-3, sum_rtol=1e-4):
        """max-abs error of the candidate against the fixture (atol + rtol*|ref|), plus the
        whole-tensor fp64 sum against sum_rtol * abs-sum.  Returns the max abs error."""
        z = self.files[file]
        shape = tuple(int(v) for v in z[f"{name}/shape"])
        c = cand.detach().to(torch.float32).cpu().contiguous().numpy()
        assert tuple(c.shape) == shape, f"{name}: shape {tuple(c.shape)} != golden {shape}"
        ref, stride = self.expected(file, name)
        got = c if stride is None else c.reshape(-1)[::stride]
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        bound = atol + rtol * np.abs(ref.astype(np.float64))
        worst = float((err - bound).max())
        assert np.isfinite(c).all(), f"{name}: non-finite values"
        assert worst <= 0, f"{name}: max|err|={err.max():.3e} exceeds atol={atol} rtol={rtol} (ref scale {np.abs(ref).max():.3e})"
        s, a = float(z[f"{name}/sum"]), float(z[f"{name}/abssum"])
        cs = float(c.astype(np.float64).sum())
        assert abs(cs - s) <= sum_rtol * a + 1e-6, f"{name}: sum {cs} vs golden {s} (abs-sum {a})"
        return float(err.max())
