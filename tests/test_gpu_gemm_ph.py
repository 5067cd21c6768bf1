"""The phase-structured NT kernels (csrc/gemm_ph.h: 256 x 256 tiles, csrc/gemm_ph3.h: 192 x 256 tiles) through fmmt_linear_fwd: whole-output race screens.

Both kernels overlap their epilogue with the next tile's K steps; a first version re-used the data registers of its output stores too early and
corrupted ~1e-5 of the elements, different ones in every launch -- far too few for a sampled comparison to notice reliably.  So: the WHOLE output
of several launches against an fp32 product of the same bf16 operands, on shapes that take each kernel by each of its dispatch rules (16384+ tokens
with full tile rounds; 4096-16383 tokens from 150 tiles; the tile shape whose rounds waste less), with and without bias, ragged last panel, and the
192 x 256 kernel's other epilogues: residual + DropPath row scale, row scale alone, GELU + pre-activation (K >= 1536)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

PLAIN = [(31360, 2304, 768), (31360 + 16, 3072, 768), (20000, 1536, 384), (31360, 768, 3072), (31360 + 8, 768, 768), (7840, 6144, 1536), (7848, 1536, 1536),
         (7840, 1536, 6144), (125440, 1536, 384), (125440, 1152, 384), (62720 + 8, 384, 384)]


def _operands(M, N, K, dev):
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    x = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    b = torch.rand(N, device=dev, generator=g) - 0.5
    return x, w, b, g


@pytest.mark.parametrize("M,N,K", PLAIN)
def test_plain_whole_output_over_several_launches(M, N, K):
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    x, w, b, _ = _operands(M, N, K, dev)
    for bias in (b, None):
        ref = x.float() @ w.float().t()
        if bias is not None:
            ref += bias
        tol = 0.01 * ref.abs() + 6e-3
        for launch in range(4):
            y = ops.linear_raw(x, w, bias)
            bad = int((~((y.float() - ref).abs() <= tol)).sum())
            assert bad == 0, (launch, bias is not None, bad)


@pytest.mark.parametrize("M,N,K", [(31360, 768, 3072), (31360 + 8, 768, 768), (7840, 1536, 6144), (7848, 1536, 1536), (31360, 3072, 768), (125440, 384, 384), (62720 + 8, 1152, 384)])
def test_residual_and_row_scale_whole_output(M, N, K):
    """y = res + rowscale[row / rows_per_scale] * (x W^T + b) (proj / fc2 forward with DropPath), and the row scale alone (proj's input gradient)"""
    from facialmmt_amd import ops
    dev = torch.device("cuda:0")
    x, w, b, g = _operands(M, N, K, dev)
    rps = 49 if M < 16384 else 196
    res = (torch.rand(M, N, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    rs = torch.rand(M // rps + 1, device=dev, generator=g) + 0.5
    rs[1] = 0.0                                                  # a dropped sample
    pre = x.float() @ w.float().t()
    srow = rs.repeat_interleave(rps)[:M, None]
    for (bias, r, name) in ((b, res, "res+scale"), (None, None, "scale"), (b, res, "res")):
        scale = None if name == "res" else rs
        ref = (pre + (bias if bias is not None else 0.0)) * (srow if scale is not None else 1.0) + (r.float() if r is not None else 0.0)
        tol = 0.01 * ref.abs() + 1.2e-2                          # the branch value is rounded to bf16 before the residual is added (one more rounding than the fp32 reference)
        for launch in range(3):
            y = ops.linear_raw(x, w, bias, res=r, rowscale=scale, rows_per_scale=rps)
            bad = int((~((y.float() - ref).abs() <= tol)).sum())
            assert bad == 0, (name, launch, bad)


@pytest.mark.parametrize("M,N,K", [(7840, 6144, 1536), (31360, 768, 3072), (7848, 1536, 1536)])
def test_gelu_with_pre_activation_whole_output(M, N, K):
    from facialmmt_amd import ops
    from facialmmt_amd._lib import EPI_GELU
    dev = torch.device("cuda:0")
    x, w, b, _ = _operands(M, N, K, dev)
    ref = x.float() @ w.float().t() + b
    gref = torch.nn.functional.gelu(ref)
    for launch in range(3):
        pre = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
        y = ops.linear_raw(x, w, b, epi=EPI_GELU, y_pre=pre)
        bad_pre = int((~((pre.float() - ref).abs() <= 0.01 * ref.abs() + 6e-3)).sum())
        bad = int((~((y.float() - gref).abs() <= 0.01 * gref.abs() + 6e-3)).sum())
        assert bad_pre == 0 and bad == 0, (launch, bad_pre, bad)
