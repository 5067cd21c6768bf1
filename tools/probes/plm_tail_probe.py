"""fmmt_plm_dropadd_ln_fwd / _bwd (+ its reduction) at the text encoder's shape (2048 tokens x 1024, p = 0.1), per launch set.  PROBE_LIB selects a build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
lib = _lib.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, C in ((2048, 1024), (2048, 768), (640, 768)):
    h = torch.randn(M, C, device=dev).to(torch.bfloat16); res = torch.randn(M, C, device=dev).to(torch.bfloat16)
    g = torch.ones(C, device=dev, dtype=torch.bfloat16); b = torch.zeros(C, device=dev, dtype=torch.bfloat16)
    xs, y, dx, dh = (torch.empty_like(h) for _ in range(4))
    dg, db, dbias = (torch.empty_like(g) for _ in range(3))
    seed = torch.tensor([3], device=dev, dtype=torch.int64)
    nb = lib.fmmt_plm_dropadd_ln_bwd_workspace(M, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for p in (0.1, 0.0):
        f = lambda: _lib.check(lib.fmmt_plm_dropadd_ln_fwd(M, C, 1e-5, h.data_ptr(), res.data_ptr(), g.data_ptr(), b.data_ptr(), p, 0, seed.data_ptr(), 5 << 40, xs.data_ptr(), y.data_ptr(), st), "f")
        bw = lambda: _lib.check(lib.fmmt_plm_dropadd_ln_bwd(M, C, 1e-5, y.data_ptr(), xs.data_ptr(), g.data_ptr(), p, 0, seed.data_ptr(), 5 << 40, dx.data_ptr(), dh.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                            dbias.data_ptr(), ws.data_ptr(), nb, st), "b")
        print(f"M={M} C={C} p={p}: fwd {timeit(f):6.1f} us   bwd + reduce {timeit(bw):6.1f} us", flush=True)

# the intermediate's backward: d(pre) = d(act) gelu'(pre) + colsum, against torch's two launches
for M, H in ((2048, 4096), (2048, 3072)):
    pre = (2.0 * torch.randn(M, H, device=dev)).to(torch.bfloat16); dact = torch.randn(M, H, device=dev).to(torch.bfloat16)
    dpre = torch.empty_like(pre); db = torch.empty(H, device=dev, dtype=torch.bfloat16)
    nb = lib.fmmt_plm_gelu_bwd_colsum_workspace(M, H); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.fmmt_plm_gelu_bwd_colsum(M, H, dact.data_ptr(), pre.data_ptr(), dpre.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, st), "g")
    x = pre.clone().requires_grad_(True); y = torch.nn.functional.gelu(x)
    def ref():
        (g,) = torch.autograd.grad(y, x, dact, retain_graph=True)
        return g.sum(0)
    print(f"M={M} H={H}: gelu' * dact + colsum {timeit(f):6.1f} us   (torch GeluBackward + sum(0): {timeit(ref):6.1f} us)", flush=True)
