#!/bin/bash
# C2: serialised kernel census of the eager engine + kernel trace of the replayed linear graph (kernel time against wall time)
bash scratch/serial_census.sh ser_c2 train_c2 > gpurun_out/ser_c2_head.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trg
BUCTD_TUNING=1 BUCTD_BRANCH_STREAMS=0 BUCTD_WGRAD_STREAM=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o t -- python $GRAFT_REPO_ROOT/scratch/step_graph_probe.py --workload train_c2 --steps 10 > /tmp/trg.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trg -name "*kernel_trace.csv" | head -1)
python - $f <<'PY' > gpurun_out/sg_linear_trace.txt 2>&1
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 10 "graph" steps = the tail of the trace: take the last 14450*... simply the last 20 % of kernels
n = len(rows)
tail = rows[int(n * 0.85):]
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail, tail[1:])]
gaps_pos = [g for g in gaps if g > 0]
print(f"kernels {len(tail)} wall {(t1 - t0) / 1e6:.2f} ms busy {busy / 1e6:.2f} ms gaps>0 {len(gaps_pos)} sum {sum(gaps_pos) / 1e6:.2f} ms median {sorted(gaps_pos)[len(gaps_pos) // 2] / 1e3:.2f} us")
PY
tail -3 /tmp/trg.log >> gpurun_out/sg_linear_trace.txt
head -12 gpurun_out/ser_c2_head.txt; cat gpurun_out/sg_linear_trace.txt
