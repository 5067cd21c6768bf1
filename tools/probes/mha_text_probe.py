"""Development aid: the text encoder's self-attention shape (RoBERTa-large: T = 512, B = 4, 16 heads of 64) on the in-tree MHA kernels
(time-major operands, packed qkv) beside torch's scaled_dot_product_attention (the kernel the HF module runs), forward and backward."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
dev = torch.device("cuda:0")


def ev(fn, n=10, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


for (T, B, H) in ((512, 4, 16), (512, 8, 16), (256, 4, 16)):
    E = H * 64
    qkv = torch.randn(T, B, 3 * E, device=dev, dtype=torch.bfloat16)
    q = qkv[..., :E].contiguous()
    kv = qkv[..., E:].contiguous()
    kb = torch.zeros(B, T, device=dev)
    kb[:, T - 37:] = -1e30
    seed = torch.randint(0, 2 ** 62, (1,), device=dev, dtype=torch.int64)
    for p in (0.0, 0.1):
        out, lse = ops.mha_fwd_raw(q, kv, None, H, 0.125, p, 0, seed, kb)
        do = torch.randn_like(out)
        tf = ev(lambda: ops.mha_fwd_raw(q, kv, None, H, 0.125, p, 0, seed, kb))
        tb = ev(lambda: ops.mha_bwd_raw(q, kv, None, out, do, lse, H, 0.125, p, 0, seed, kb))
        q4 = torch.randn(B, H, T, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
        k4 = torch.randn(B, H, T, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
        v4 = torch.randn(B, H, T, 64, device=dev, dtype=torch.bfloat16, requires_grad=True)
        m4 = kb.view(B, 1, 1, T).to(torch.bfloat16).expand(B, 1, T, T)
        sd = lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, attn_mask=m4, dropout_p=p)
        o4 = sd()
        g4 = torch.randn_like(o4)
        tsf = ev(sd)
        tsb = ev(lambda: torch.autograd.grad(sd(), [q4, k4, v4], g4)) - tsf
        fl = 4.0 * B * H * T * T * 64
        print(f"T={T} B={B} H={H} p={p}: in-tree fwd {tf:7.1f} us ({fl/tf/1e6:6.1f} TF/s) bwd {tb:7.1f} us ({2.5*fl/tb/1e6:6.1f} TF/s) | sdpa fwd {tsf:7.1f} us bwd {tsb:7.1f} us", flush=True)
