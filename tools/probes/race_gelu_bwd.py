"""Development aid: the stage-1 / stage-2 GELU' input-gradient launches repeated, outputs compared bit for bit with the first."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import ops
from facialmmt_amd._lib import EPI_GELU_BWD, EPI_GELU
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K, epi) in [(501760, 768, 192, EPI_GELU_BWD), (125440, 1536, 384, EPI_GELU_BWD), (125440, 1536, 384, EPI_GELU), (31360, 3072, 768, EPI_GELU), (501760, 768, 192, EPI_GELU)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
    aux = torch.randn(M, N, device=dev).bfloat16() if epi == EPI_GELU_BWD else None
    rs = torch.full((M // 196,), 1.0 / 0.9, device=dev)
    first, bad = None, 0
    for it in range(150):
        if epi == EPI_GELU_BWD:
            y = ops.linear_raw(x, w, None, epi=epi, aux=aux, rowscale=rs, rows_per_scale=196)
        else:
            pre = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            y = ops.linear_raw(x, w, torch.zeros(N, device=dev), epi=epi, y_pre=pre)
        if first is None:
            first = y.clone()
        elif not torch.equal(first, y):
            bad += 1
            d = (first.float() - y.float()).abs()
            rows = torch.nonzero(d.amax(1) > 0).flatten()
            r0 = int(rows[0]); cols = torch.nonzero(d[r0] > 0).flatten()[:6]
            gp = lambda t: (0.5 * (1 + torch.erf(t / 2 ** 0.5)) + t * torch.exp(-0.5 * t * t) * 0.3989422804)
            print("   right", first[r0, cols].float().tolist(), "\n   wrong", y[r0, cols].float().tolist(), "\n   aux  ", aux[r0, cols].float().tolist(),
                  "\n   gelu'(aux)", gp(aux[r0, cols].float()).tolist(), "\n   implied gelu' wrong", (y[r0, cols].float() / first[r0, cols].float() * gp(aux[r0, cols].float())).tolist(), flush=True)
            print(f"  it {it}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}, rows {rows[:6].tolist()} .. ({len(rows)} rows), cols of first row {torch.nonzero(d[rows[0]] > 0).flatten()[:8].tolist()}", flush=True)
    print(f"{M}x{N}x{K} epi {epi}: {bad} of 149 repeats differ", flush=True)
