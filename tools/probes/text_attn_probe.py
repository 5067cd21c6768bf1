"""The text encoder's attention (RoBERTa-large / BERT-large: 16 heads x 64, 4 x 512 tokens, attention dropout 0.1) on fmmt_mha_fwd / fmmt_mha_bwd against
torch's scaled_dot_product_attention (what transformers' sdpa interface runs), forward and backward, per launch set.  PROBE_LIB selects a build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, n=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for (B, S, H, D) in ((4, 512, 16, 64), (1, 512, 16, 64), (4, 128, 16, 64)):
    E = H * D
    p = 0.1
    # ours: time-major packed (S, B, 3E)
    qkv_t = torch.randn(S, B, 3 * E, device=dev).to(torch.bfloat16)
    q, k, v = qkv_t[..., :E].contiguous(), qkv_t[..., E:2 * E].contiguous(), qkv_t[..., 2 * E:].contiguous()
    seed = torch.tensor([5], device=dev, dtype=torch.int64)
    scale = D ** -0.5
    out, lse = ops.mha_fwd_raw(q, k, v, H, scale, p, 0, seed, None)
    dout = torch.randn_like(out)
    t_f = timeit(lambda: ops.mha_fwd_raw(q, k, v, H, scale, p, 0, seed, None))
    t_b = timeit(lambda: ops.mha_bwd_raw(q, k, v, out, dout, lse, H, scale, p, 0, seed, None))
    t_f0 = timeit(lambda: ops.mha_fwd_raw(q, k, v, H, scale, 0.0, 0, seed, None))
    t_b0 = timeit(lambda: ops.mha_bwd_raw(q, k, v, out, dout, lse, H, scale, 0.0, 0, seed, None))
    # torch: batch-major packed (B, S, 3E), heads as transposed views (what transformers hands sdpa)
    qkv_b = torch.randn(B, S, 3 * E, device=dev).to(torch.bfloat16).requires_grad_(True)
    def heads(i):
        return qkv_b[..., i * E:(i + 1) * E].view(B, S, H, D).transpose(1, 2)
    def sdpa(pp):
        return torch.nn.functional.scaled_dot_product_attention(heads(0), heads(1), heads(2), attn_mask=None, dropout_p=pp, scale=scale)
    o = sdpa(p)
    do = torch.randn_like(o)
    s_f = timeit(lambda: sdpa(p))
    def fb(pp):
        qkv_b.grad = None
        sdpa(pp).backward(do)
    s_fb = timeit(lambda: fb(p))
    s_f0 = timeit(lambda: sdpa(0.0))
    s_fb0 = timeit(lambda: fb(0.0))
    fl = 4.0 * B * H * S * S * D
    print(f"B={B} S={S} H={H} D={D}: fmmt fwd {t_f:6.1f} us ({fl / t_f * 1e-6:5.1f} TF/s) bwd {t_b:6.1f} us | p=0: fwd {t_f0:6.1f} bwd {t_b0:6.1f} || "
          f"torch sdpa fwd {s_f:6.1f} us, fwd+bwd {s_fb:6.1f} us | p=0: fwd {s_f0:6.1f} fwd+bwd {s_fb0:6.1f}", flush=True)

# batch-major entry against the time-major one on the transposed operands: same bits (layout only)
for (B, S, H, D, p) in ((4, 512, 16, 64, 0.1), (3, 200, 12, 64, 0.0), (2, 77, 16, 64, 0.2)):
    E = H * D
    scale = D ** -0.5
    qkv = torch.randn(B, S, 3 * E, device=dev).to(torch.bfloat16)
    kbias = torch.zeros(B, S, device=dev)
    kbias[:, S - 5:] = -10000.0
    seed = torch.tensor([11], device=dev, dtype=torch.int64)
    for kb in (None, kbias):
        out, lse = ops.mha_packed_bm_fwd_raw(qkv, H, scale, p, 0, seed, kb)
        dout = torch.randn_like(out)
        dqkv = ops.mha_packed_bm_bwd_raw(qkv, out, dout, lse, H, scale, p, 0, seed, kb)
        t = qkv.transpose(0, 1).contiguous()
        q, k, v = t[..., :E].contiguous(), t[..., E:2 * E].contiguous(), t[..., 2 * E:].contiguous()
        o2, l2 = ops.mha_fwd_raw(q, k, v, H, scale, p, 0, seed, kb)
        dq, dk, dv = ops.mha_bwd_raw(q, k, v, o2, dout.transpose(0, 1).contiguous(), l2, H, scale, p, 0, seed, kb)
        d2 = torch.cat([dq, dk, dv], dim=-1).transpose(0, 1)
        print(f"batch-major B={B} S={S} H={H} p={p} key_bias={'yes' if kb is not None else 'no'}: out equal {torch.equal(out, o2.transpose(0, 1))}, lse equal {torch.equal(lse, l2)}, "
              f"dqkv equal {torch.equal(dqkv, d2)}", flush=True)
        # against torch (p = 0)
        if p == 0.0:
            qh = qkv.float().view(B, S, 3, H, D).permute(2, 0, 3, 1, 4).requires_grad_(True)
            am = kb[:, None, None, :] if kb is not None else None
            ref = torch.nn.functional.scaled_dot_product_attention(qh[0], qh[1], qh[2], attn_mask=am, scale=scale)
            ref2 = ref.transpose(1, 2).reshape(B, S, E)
            (g,) = torch.autograd.grad(ref2, qh, dout.float())
            gref = g.permute(1, 3, 0, 2, 4).reshape(B, S, 3 * E)
            print(f"    vs torch fp32: out {(out.float() - ref2).abs().max().item():.3e} (scale {ref2.abs().max().item():.2f}), dqkv {(dqkv.float() - gref).abs().max().item():.3e} "
                  f"(scale {gref.abs().max().item():.2f})", flush=True)
