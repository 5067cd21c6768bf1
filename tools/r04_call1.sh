#!/bin/bash
# round-4 GPU call 1: suite + A/B probes (base = round-3 kernels, prio = s_setprio for the younger half) + one bench line
O=gpurun_out/r04; mkdir -p $O
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
B=$PWD/facialmmt_amd/libfmmt_hip_base.so; P=$PWD/facialmmt_amd/libfmmt_hip_prio.so
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -x -p no:cacheprovider > $O/pytest1.log 2>&1; echo "pytest rc=$?" >> $O/pytest1.log
for v in base new; do
  L=""; [ $v = base ] && L=$B
  PROBE_LIB=$L timeout 300 python tools/probes/wattn_bench.py > $O/wattn_$v.txt 2>&1
done
for v in base new prio; do
  L=""; [ $v = base ] && L=$B; [ $v = prio ] && L=$P
  PROBE_LIB=$L timeout 300 python tools/probes/nt_probe.py > $O/nt_$v.txt 2>&1
  PROBE_LIB=$L timeout 300 python tools/probes/time_swin.py 640 > $O/swin_$v.txt 2>&1
done
PROBE_LIB=$P timeout 300 python tools/probes/mlp_bench.py > $O/mlp_prio.txt 2>&1
timeout 300 python tools/probes/mlp_bench.py > $O/mlp_new.txt 2>&1
timeout 120 python - > $O/profiler_smoke.txt 2>&1 <<'PY'
import torch
from torch.profiler import profile, ProfilerActivity
x = torch.randn(4096, 4096, device="cuda")
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        y = x @ x
    torch.cuda.synchronize()
n = 0
for ev in prof.events():
    if str(getattr(ev, "device_type", "")).endswith("CUDA"):
        n += 1
        if n < 6: print(ev.name[:60], ev.device_type, ev.device_time_total)
print("cuda events", n)
PY
timeout 900 python bench.py --other-configs 0 --host-input-leg 0 > $O/bench1.json 2> $O/bench1.err
tail -c 600 $O/bench1.err
