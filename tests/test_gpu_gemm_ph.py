"""The four-phase NT kernel (csrc/gemm_ph.h) through fmmt_linear_fwd: a whole-output race screen.

The kernel overlaps its epilogue with the next tile's K steps; a first version re-used the data registers of its output stores too early and
corrupted ~1e-5 of the elements, different ones in every launch -- far too few for a sampled comparison to notice reliably.  So: the WHOLE output
of several launches against an fp32 product of the same bf16 operands, on shapes that take the kernel by both of its dispatch rules (16384+ tokens
with full tile rounds; 4096-16383 tokens from 150 tiles), with and without bias, ragged last panel; the GELU + pre-activation launches of the
same shapes (on the persistent / 128-row kernels while FMMT_NT_PH_GELU = 0) are held to the same whole-output bar."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,gelu", [(31360, 2304, 768, False), (31360 + 16, 3072, 768, False), (20000, 1536, 384, False),
                                       (7840, 6144, 1536, True), (7848, 1536, 1536, True), (7840, 1536, 6144, False)])
def test_whole_output_over_several_launches(M, N, K, gelu):
    from facialmmt_amd import ops
    from facialmmt_amd._lib import EPI_GELU
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    x = (torch.rand(M, K, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device=dev, generator=g) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    b = torch.rand(N, device=dev, generator=g) - 0.5
    for bias in (b, None):
        ref = x.float() @ w.float().t()
        if bias is not None:
            ref += bias
        tol = 0.01 * ref.abs() + 6e-3
        for launch in range(4):
            if gelu:
                pre = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
                y = ops.linear_raw(x, w, bias, epi=EPI_GELU, y_pre=pre)
                bad_pre = int(((pre.float() - ref).abs() > tol).sum())
                gref = torch.nn.functional.gelu(ref)
                bad = int(((y.float() - gref).abs() > 0.01 * gref.abs() + 8e-3).sum())
                assert bad_pre == 0 and bad == 0, (launch, bias is not None, bad_pre, bad)
            else:
                y = ops.linear_raw(x, w, bias)
                bad = int((~((y.float() - ref).abs() <= tol)).sum())
                assert bad == 0, (launch, bias is not None, bad)
