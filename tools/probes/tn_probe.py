"""Development aid: the stage-2/3 weight-gradient launches alone, timed with HIP events (A/B of the FMMT_TN_* switches: one
process per setting, same gpurun call), or bare for PMC passes (rocprofv3 --pmc ... -- python tools/probes/tn_probe.py --bare)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from facialmmt_amd import _lib
if os.environ.get("PROBE_LIB"):                                # A/B of two builds in one gpurun call
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from facialmmt_amd import ops
dev = torch.device("cuda:0")
bare = "--bare" in sys.argv
SHAPES = [(125440, 1536, 384), (125440, 1152, 384), (125440, 384, 1536), (125440, 384, 384), (125440, 384, 768),
          (31360, 2304, 768), (31360, 3072, 768), (31360, 768, 3072), (31360, 768, 768), (31360, 768, 1536)]
if bare:
    SHAPES = SHAPES[:3]
tot = 0.0
for (M, N, K) in SHAPES:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16); x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for _ in range(4):
        ops.wgrad_raw(dy, x, True)
    torch.cuda.synchronize()
    if bare:
        continue
    from facialmmt_amd import _lib
    nbytes = _lib.load().fmmt_linear_wgrad_workspace(ops.dtype_code(dy.dtype), M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.wgrad_partials_raw(dy, x, True, ws, nbytes)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.wgrad_raw(dy, x, True)
    e1.record(); torch.cuda.synchronize()
    full = e0.elapsed_time(e1) / 10
    tot += full
    rps = 196 if M > 100000 else 49
    rs = torch.full((M // rps,), 1.0 / 0.9, device=dev)
    if int(os.environ.get('TN_PROBE_DROP', '10')):
        rs[::int(os.environ.get('TN_PROBE_DROP', '10'))] = 0.0    # DropPath's largest rate in Swin-T (0.1): one image in ten dropped (TN_PROBE_DROP=0: none)
    ops.wgrad_partials_raw(dy, x, True, ws, nbytes, rs, rps); torch.cuda.synchronize()
    bs = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.wgrad_partials_raw(dy, x, True, ws, nbytes, rs, rps)
        e1.record(); torch.cuda.synchronize()
        bs = min(bs, e0.elapsed_time(e1) / 10)
    tot += bs
    print(f"  tn {M:7d}x{N:5d}x{K:5d}: partials {best*1e3:7.1f} us {2.0*M*N*K/best/1e9:6.1f} TF/s | with reduce {full*1e3:7.1f} us | DropPath-scaled partials {bs*1e3:7.1f} us {2.0*M*N*K/bs/1e9:6.1f} TF/s", flush=True)
if not bare:
    print(f"  total with reduce {tot*1e3:.1f} us   FMMT_TN_DMA={os.environ.get('FMMT_TN_DMA', '')}")
