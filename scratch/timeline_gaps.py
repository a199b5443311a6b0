"""GPU-idle accounting of the bench from a rocprofv3 --kernel-trace CSV: python scratch/timeline_gaps.py <kernel_trace.csv> [steps]
Splits the trace at the Adam kernel (one per step), drops warm-up, and reports per step: wall time, time with at least one
kernel running, per-queue busy time, and the idle gaps (no kernel on any queue)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
print(f"{len(ks)} kernels, {len(adam)} adam launches")
steps = []
for a, b in zip(adam[3:-1], adam[4:]):          # skip warm-up
    steps.append(ks[a + 1:b + 1])
steps = steps[:int(sys.argv[2]) if len(sys.argv) > 2 else 8]
for si, st in enumerate(steps):
    t0, t1 = st[0][0], max(k[1] for k in st)
    ev = sorted([(k[0], 1) for k in st] + [(k[1], -1) for k in st])
    busy = 0; depth = 0; last = t0; gaps = []; conc = collections.Counter()
    for t, d in ev:
        if depth > 0: busy += t - last
        elif t - last > 0: gaps.append((t - last, last - t0))
        conc[depth] += t - last
        depth += d; last = t
    perq = collections.Counter()
    for k in st: perq[k[3]] += k[1] - k[0]
    big = [g for g in gaps if g[0] > 3000]
    print(f"step {si}: wall {(t1-t0)/1e6:.2f} ms, any-kernel busy {busy/1e6:.2f} ms, idle {sum(g[0] for g in gaps)/1e6:.2f} ms in {len(gaps)} gaps "
          f"({len(big)} gaps > 3 us = {sum(g[0] for g in big)/1e6:.2f} ms); sum of kernel durations {sum(k[1]-k[0] for k in st)/1e6:.2f} ms; "
          f"time at concurrency 1/2/3/4+: {conc[1]/1e6:.1f}/{conc[2]/1e6:.1f}/{conc[3]/1e6:.1f}/{sum(v for c, v in conc.items() if c >= 4)/1e6:.1f} ms; "
          f"per queue busy: {', '.join(f'{q}:{v/1e6:.1f}' for q, v in sorted(perq.items()))}")
    if si == 0:
        # where in the step are the big gaps (ms offsets)
        print("   largest gaps (us @ ms):", [f"{g[0]/1e3:.0f}@{g[1]/1e6:.1f}" for g in sorted(gaps, reverse=True)[:15]])

# who runs alone: attribute the time at concurrency 1 to the kernel that is running then (step 1)
st = steps[1] if len(steps) > 1 else steps[0]
ev = sorted([(k[0], 1, i) for i, k in enumerate(st)] + [(k[1], -1, i) for i, k in enumerate(st)])
active = set(); last = st[0][0]; alone = collections.Counter(); alone_n = collections.Counter(); total = collections.Counter(); grid = {}
for t, d, i in ev:
    if len(active) == 1:
        j = next(iter(active)); alone[st[j][2][:70]] += t - last
    last = t
    if d == 1: active.add(i)
    else: active.discard(i)
for k in st:
    total[k[2][:70]] += k[1] - k[0]; alone_n[k[2][:70]] += 1
print("time running ALONE on the GPU, by kernel (step 1):")
for name, v in alone.most_common(28):
    print(f"  {v/1e6:6.2f} ms alone of {total[name]/1e6:6.2f} ms total, {alone_n[name]:4d} launches  {name}")
