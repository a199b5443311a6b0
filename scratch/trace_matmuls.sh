# every matmul_kernel / x6_gemm / x6_image / splitk launch of one traced C4 step with grid and duration
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trm
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trm -o t -- python $GRAFT_REPO_ROOT/bench.py --workload train_c4 --steps 4 --warmup 3 --no-cpu-baseline --no-kernel-timer > /tmp/trm.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trm -name "*kernel_trace.csv" | head -1)
python - $f <<'PY' > gpurun_out/matmul_launches.txt 2>&1
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
a, b = adam[-2], adam[-1]
for r in rows[a:b]:
    n = r["Kernel_Name"]
    if any(k in n for k in ("matmul", "x6_gemm", "x6_image", "splitk", "softmax", "colsum_part")):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:9.1f} us  grid {r['Grid_Size_X']:>8} {r['Grid_Size_Y']:>5} {r['Grid_Size_Z']:>5}  wg {r['Workgroup_Size_X']:>4}  {n[:90]}")
PY
cat gpurun_out/matmul_launches.txt
