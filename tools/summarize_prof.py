"""Turn rocprofv3 CSV output (kernel_stats / kernel_trace / counter_collection) into the compact
summaries committed under profiles/.  Usage:
    python tools/summarize_prof.py stats <dir> <n_steps_profiled> > profiles/rNN_kernel_stats.md
    python tools/summarize_prof.py pmc   <dir> > profiles/rNN_pmc_<counter>.md
"""
import collections
import csv
import glob
import os
import sys


def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not m:
        sys.exit(f"no *{suffix} under {d}")
    return m[0]


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    return name[:110]


def stats(d, steps):
    rows = list(csv.DictReader(open(find(d, "kernel_stats.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats: per-kernel GPU time ({steps} steps incl. warm-up profiled)\n")
    print(f"total kernel time: {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.2f} ms/step\n")
    print("| ms/step | % | calls/step | avg us | min us | max us | kernel |\n|---:|---:|---:|---:|---:|---:|---|")
    for r in rows[:int(os.environ.get("PROF_ROWS", "45"))]:
        print(f"| {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {float(r['Percentage']):.2f} | {int(r['Calls']) / steps:.1f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | `{short(r['Name'])}` |")


def pmc(d):
    f = find(d, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    print("# rocprofv3 --pmc (own pass): per-kernel counter sums and per-dispatch averages\n")
    print("| kernel | counter | dispatches | sum | avg / dispatch |\n|---|---|---:|---:|---:|")
    order = sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values()))
    for k, cs in order[:int(os.environ.get("PMC_ROWS", "60"))]:
        for c, (n, s) in sorted(cs.items()):
            print(f"| `{k[:90]}` | {c} | {n} | {s:.4g} | {s / n:.4g} |")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    else:
        pmc(sys.argv[2])
