"""Timing of the uint8 pre-step (fmmt_patch_embed_u8_ln_fwd, training and inference form; fmmt_patch_embed_u8 alone) at the bench geometry (640 crops of 112 x 112,
and 160 x 160 / 224 x 224 for the table's other sizes).  PROBE_LIB=... for another build."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6


g = torch.Generator(device=dev).manual_seed(1)
w = (torch.randn(96, 48, device=dev) * 0.1).requires_grad_(True)
b = torch.zeros(96, device=dev, requires_grad=True)
gam = torch.ones(96, device=dev, requires_grad=True); bet = torch.zeros(96, device=dev, requires_grad=True)
for S in (112, 160, 224):
    img = torch.randint(0, 256, (640, S, S, 3), generator=g, device=dev, dtype=torch.uint8)
    for mode in ("pil", "cv2"):
        t_train = timeit(lambda: ops.patch_embed_u8_ln(img, mode, w, b, gam, bet, 1e-5, torch.bfloat16))
        with torch.no_grad():
            t_inf = timeit(lambda: ops.patch_embed_u8_ln(img, mode, w, b, gam, bet, 1e-5, torch.bfloat16))
        t_cols = timeit(lambda: ops.patch_embed_u8(img, mode, torch.bfloat16))
        byts = 640 * (S * S * 3 + 3136 * (96 * 2 * 2 + 48 * 2 + 8))
        print(f"u8 pre-step S={S} {mode}: fused training {t_train:.1f} us ({byts / t_train / 1e3:.0f} GB/s algorithmic) | fused inference {t_inf:.1f} us | patch matrix alone {t_cols:.1f} us", flush=True)
