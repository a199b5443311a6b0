"""Last-arrival critical path of one train step from a rocprofv3 --kernel-trace CSV:
    python scratch/critical_path.py <kernel_trace.csv> [step index]
Walk back from the step's last kernel: the predecessor of a kernel is the kernel (any queue) that ended last before it
started - the one whose completion (stream order or a cross-stream event) released it, or, if the gap to it is long, the
host.  Reports the chain's time by kernel name, the gaps on it, and how often the chain changes queues."""
import csv, sys, collections, bisect
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows), key=lambda t: t[0])
adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
si = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = adam[3 + si], adam[4 + si]
st = ks[a + 1:b + 1]
by_end = sorted(range(len(st)), key=lambda i: st[i][1])
ends = [st[i][1] for i in by_end]
cur = max(range(len(st)), key=lambda i: st[i][1])
t_end = st[cur][1]
chain_time = collections.Counter(); chain_n = collections.Counter(); gaps = []; switches = 0; nk = 0
while True:
    s, e, name, q = st[cur]
    chain_time[name[:72]] += e - s; chain_n[name[:72]] += 1; nk += 1
    # latest-ending kernel with end <= start + 1 us
    j = bisect.bisect_right(ends, s + 1000) - 1
    while j >= 0 and by_end[j] == cur: j -= 1
    if j < 0: break
    pred = by_end[j]
    gap = s - st[pred][1]
    gaps.append((max(gap, 0), name[:50], st[pred][2][:50]))
    if st[pred][3] != q: switches += 1
    cur = pred
    if st[cur][0] <= st[0][0] and cur == 0: 
        chain_time[st[cur][2][:72]] += st[cur][1] - st[cur][0]; chain_n[st[cur][2][:72]] += 1; nk += 1
        break
wall = t_end - st[0][0]
tot = sum(chain_time.values()); g = sum(x[0] for x in gaps)
print(f"step {si}: wall {wall/1e6:.2f} ms; critical chain: {nk} of {len(st)} kernels, {tot/1e6:.2f} ms in kernels + {g/1e6:.2f} ms in {len(gaps)} gaps "
      f"({sum(1 for x in gaps if x[0] > 3000)} gaps > 3 us = {sum(x[0] for x in gaps if x[0] > 3000)/1e6:.2f} ms), {switches} queue changes")
alltime = collections.Counter(); alln = collections.Counter()
for k in st: alltime[k[2][:72]] += k[1] - k[0]; alln[k[2][:72]] += 1
print("   on-chain ms / launches   (whole step ms / launches)   kernel")
for name, v in chain_time.most_common(32):
    print(f"  {v/1e6:6.2f} {chain_n[name]:5d}   ({alltime[name]/1e6:6.2f} {alln[name]:5d})   {name}")
print("largest gaps on the chain (us: kernel <- predecessor):")
for gp, nm, pn in sorted(gaps, reverse=True)[:12]:
    print(f"  {gp/1e3:7.1f}  {nm}  <-  {pn}")

# per-queue activity: when does each HIP queue finish its part of the step, and who is busy during the last 10 ms?
print("per queue: first start / last end (ms from step start), busy ms, busy in the last 10 ms before the step's last kernel ends:")
qs = collections.defaultdict(list)
for k in st: qs[k[3]].append(k)
t0 = st[0][0]
for q, lst in sorted(qs.items()):
    lastwin = sum(max(0, min(k[1], t_end) - max(k[0], t_end - 10_000_000)) for k in lst if k[1] > t_end - 10_000_000)
    names = collections.Counter()
    for k in lst:
        if k[1] > t_end - 10_000_000: names[k[2][:40]] += max(0, min(k[1], t_end) - max(k[0], t_end - 10_000_000))
    top = ", ".join(f"{n} {v/1e6:.1f}" for n, v in names.most_common(3))
    print(f"  queue {q}: {(lst[0][0]-t0)/1e6:6.2f} .. {(max(k[1] for k in lst)-t0)/1e6:6.2f}  busy {sum(k[1]-k[0] for k in lst)/1e6:6.2f}  last-10ms busy {lastwin/1e6:5.2f}  [{top}]")
