"""Timeline of one steady-state training step from a rocprofv3 --kernel-trace CSV: wall, busy (union of kernel intervals),
sum of kernel durations, how much of the wall has 0 / 1 / >= 2 kernels in flight, per-queue busy time, and the kernels that
run ALONE for the longest total time (critical-path candidates).  usage: timeline.py DIR [marker-kernel-substring]"""
import csv, glob, sys, collections

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "patch_embed_u8"
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
marks = [s for s, e, n, q, st in rows if marker in n]
print(f"{len(rows)} kernels, {len(marks)} markers")
if len(marks) < 6:
    sys.exit("not enough markers")
k = len(marks) - 4                                   # a late replayed step (before the eager probe passes: pick by regular spacing)
gaps = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
med = sorted(gaps)[len(gaps) // 2]
cands = [i for i in range(len(gaps)) if abs(gaps[i] - med) < 0.05 * med]
i = cands[len(cands) // 2]
t0, t1 = marks[i], marks[i + 1]
step = [(s, e, n, q, st) for s, e, n, q, st in rows if s >= t0 and s < t1]
print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(step)} kernels, median marker spacing {med / 1e6:.2f} ms")
ev = []
for s, e, n, q, st in step:
    ev.append((s, 1, n)); ev.append((min(e, t1), -1, n))
ev.sort()
active = collections.Counter(); cur = 0; last = t0
hist = collections.Counter(); alone = collections.Counter()
for t, dlt, n in ev:
    if t > last:
        hist[min(cur, 3)] += t - last
        if cur == 1:
            alone[[a for a, c in active.items() if c > 0][0]] += t - last
    last = t
    cur += dlt
    active[n] += dlt
hist[min(cur, 3)] += t1 - last
tot = t1 - t0
print("in flight: " + "  ".join(f"{k}{'+' if k == 3 else ''}: {v / 1e6:6.2f} ms ({100 * v / tot:4.1f} %)" for k, v in sorted(hist.items())))
print(f"sum of kernel durations {sum(e - s for s, e, *_ in step) / 1e6:.2f} ms")
byq = collections.Counter()
for s, e, n, q, st in step:
    byq[(q, st)] += e - s
print("per (queue, stream) kernel time: " + "  ".join(f"{k}: {v / 1e6:.2f} ms" for k, v in sorted(byq.items(), key=lambda kv: -kv[1])[:6]))
print("kernels running alone, total ms:")
for n, v in alone.most_common(28):
    print(f"  {v / 1e6:7.3f}  {n[:110]}")
# idle gaps
idle = []
cur = 0; last = t0
for t, dlt, n in ev:
    if cur == 0 and t > last:
        idle.append((t - last, last - t0))
    last = t; cur += dlt
idle.sort(reverse=True)
print("largest idle gaps (us @ offset ms): " + "  ".join(f"{g / 1e3:.1f}@{o / 1e6:.1f}" for g, o in idle[:12]) + f"   total idle {sum(g for g, _ in idle) / 1e6:.2f} ms in {len(idle)} gaps")
# context of the tiny launches: most common (previous kernel, next kernel) on the same queue
if len(sys.argv) > 3:
    for pat in sys.argv[3].split(","):
        ctx = collections.Counter()
        byqueue = collections.defaultdict(list)
        for s, e, n, q, st in step:
            byqueue[q].append(n)
        for q, names in byqueue.items():
            for j, n in enumerate(names):
                if pat in n:
                    pv = names[j - 1][:60] if j else "-"
                    nx = names[j + 1][:60] if j + 1 < len(names) else "-"
                    ctx[(q, pv, nx)] += 1
        print(f"context of '{pat}':")
        for (q, pv, nx), c in ctx.most_common(14):
            print(f"  {c:4d}  q{q}  after [{pv}]  before [{nx}]")
cnt = collections.Counter(); tim = collections.Counter()
for s_, e_, n, q, st in step:
    cnt[n] += 1; tim[n] += e_ - s_
print("launch census of the step (count, total ms, avg us):")
for n, c in cnt.most_common(45):
    print(f"  {c:5d} {tim[n] / 1e6:7.3f} {tim[n] / c / 1e3:7.1f}  {n[:120]}")
small = sum(c for n, c in cnt.items() if tim[n] / c < 8e3); smallt = sum(tim[n] for n, c in cnt.items() if tim[n] / c < 8e3)
print(f"kernels averaging < 8 us: {small} launches, {smallt / 1e6:.2f} ms")
# coarse phase map: per millisecond of the step and per hardware queue, the busy time and the kernel that holds most of it
print("phase map (ms offset | per queue: busy us, dominant kernel):")
qs = [q for (q, st), v in sorted(byq.items(), key=lambda kv: -kv[1])][:3]
nb = int((t1 - t0) / 1e6) + 1
bins = [{q: collections.Counter() for q in qs} for _ in range(nb)]
for s_, e_, n, q, st in step:
    if q not in qs:
        continue
    b0, b1 = int((s_ - t0) / 1e6), int((min(e_, t1) - 1 - t0) / 1e6)
    for b in range(b0, b1 + 1):
        lo, hi = max(s_, t0 + b * 1000000), min(e_, t0 + (b + 1) * 1000000)
        if hi > lo and 0 <= b < nb:
            bins[b][q][n] += hi - lo
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    for k in ("_ZN12_GLOBAL__N_1",):
        if n.startswith(k):
            n = n[len(k):].lstrip("0123456789")
    if n.startswith("at::native::") or "elementwise" in n:      # keep the functor: "vectorized_elementwise_kernel<4, CUDAFunctor_add<float>"
        import re as _re
        m = _re.search(r"(\w+Functor\w*<[^,>]*|\w+_kernel_cuda\w*|\w+KernelImpl\w*|masked_scale_kernel<[^,>]*,[^,>]*|FillFunctor<[^>]*>|\w+_copy_kernel\w*)", n)
        if m:
            return (n.split("<")[0].replace("at::native::", "")[:22] + ":" + m.group(1))[:60]
    return n[:34]
for b in range(nb):
    cells = []
    for q in qs:
        c = bins[b][q]
        busy = sum(c.values()) / 1e3
        top = short(c.most_common(1)[0][0]) if c else "-"
        cells.append(f"q{q}: {busy:5.0f} {top[:34]:34s}")
    print(f"  {b:3d} | " + " | ".join(cells))
# the launches in front of the optimizer's one kernel (hand-over, norm): name, start offset, duration, gap to the previous launch's end
ad = [i for i, r in enumerate(step) if "adamw_batch" in r[2]]
if ad:
    i1 = ad[-1]
    i0 = max(0, i1 - 70)
    print("launches in front of adamw_batch_kernel (offset ms | us | gap us | queue | kernel):")
    prev_end = None
    for s_, e_, n, q, st in step[i0:i1 + 2]:
        gap = (s_ - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"  {(s_ - t0) / 1e6:8.3f} {(e_ - s_) / 1e3:8.1f} {gap:8.1f}  q{q}  {short(n)}")
        prev_end = max(prev_end or e_, e_)
# TL_WINDOW="a,b" (ms offsets into the step): every launch that starts inside, in start order, with the idle time in front of it on ANY queue
import os as _os
if _os.environ.get("TL_WINDOW"):
    a_, b_ = [float(v) for v in _os.environ["TL_WINDOW"].split(",")]
    print(f"launches in [{a_}, {b_}) ms (offset ms | us | gap us | queue | kernel):")
    prev_end = None
    agg = collections.Counter(); cnt2 = collections.Counter()
    for s_, e_, n, q, st in step:
        off = (s_ - t0) / 1e6
        if off < a_ or off >= b_:
            if off < a_:
                prev_end = max(prev_end or e_, e_)
            continue
        gap = (s_ - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"  {off:8.3f} {(e_ - s_) / 1e3:8.1f} {max(gap, 0.0):8.1f}  q{q}  {short(n)}")
        prev_end = max(prev_end or e_, e_)
        agg[short(n)] += e_ - s_; cnt2[short(n)] += 1
    print("  by kernel (count, total us):")
    for n, v in agg.most_common(30):
        print(f"    {cnt2[n]:4d} {v / 1e3:9.1f}  {n}")
