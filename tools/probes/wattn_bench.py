"""Development micro-benchmark of the window-attention core at the bench geometry (bf16, 640 frames)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
from oracle import swin as OS
dev = torch.device("cuda:0")
N = int(os.environ.get("WATTN_FRAMES", "640"))
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tot_f = tot_b = 0.0
for (H, C, nh, blocks) in [(56, 96, 3, 2), (28, 192, 6, 2), (14, 384, 12, 6), (7, 768, 24, 2)]:
    for shift in ((0, 3) if H > 7 else (0,)):
        T = N * H * H
        qkv = torch.randn(T, 3 * C, device=dev, dtype=torch.bfloat16).requires_grad_(True)
        table = (torch.randn(169, nh, device=dev) * 0.02).requires_grad_(True)
        index = OS.relative_position_index(7).to(dev).to(torch.int32)
        mask = OS.shift_mask(H, H, 7, shift).to(dev).float() if shift else None
        fwd = lambda: ops.window_attn_core(qkv, table, index, mask, N, H, H, nh, shift, 32 ** -0.5, mask_is_shift=shift > 0)
        tf = timeit(fwd)
        out = fwd()
        dy = torch.randn_like(out)
        def bwd():
            qkv.grad = None; table.grad = None
            out.backward(dy, retain_graph=True)
        tb = timeit(bwd)
        n_launch = blocks // 2 if H > 7 else blocks
        by_f, by_b = T * C * 2 * 4, T * C * 2 * 8
        print(f"H{H:3d} C{C:4d} shift{shift}: fwd {tf*1e3:6.3f} ms {by_f/tf/1e9:6.0f} GB/s | bwd {tb*1e3:6.3f} ms {by_b/tb/1e9:6.0f} GB/s   (x{n_launch} per step)", flush=True)
        tot_f += tf * n_launch; tot_b += tb * n_launch
        del qkv, out, dy
print(f"per step: fwd {tot_f*1e3:.2f} ms, bwd {tot_b*1e3:.2f} ms")
