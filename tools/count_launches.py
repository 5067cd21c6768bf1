"""Launch census from a rocprofv3 --kernel-trace --stats directory: kernels sorted by CALLS per step (the host-side
cost of a HIP-graph replay on this ROCm version is per node, ~10 us, so launch count -- not only GPU time -- bounds the
step).  Usage: python tools/count_launches.py <dir> <steps> [rows]"""
import csv
import glob
import os
import sys

d, steps = sys.argv[1], int(sys.argv[2])
rows_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot_calls = sum(int(r["Calls"]) for r in rows)
tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# launch census: {tot_calls / steps:.0f} kernel launches/step, {tot_ns / 1e6 / steps:.2f} ms of kernel time/step ({steps} steps profiled)\n")
print("| calls/step | ms/step | avg us | kernel |\n|---:|---:|---:|---|")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:rows_n]:
    print(f"| {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['AverageNs']) / 1e3:.1f} | `{r['Name'][:120]}` |")
