"""copy the artefacts of scratch/r06_profiles.sh (gpurun_out/r06) into profiles/ under their round-6 names, with headers:
python scratch/r06_collect.py [date]"""
import os, shutil, sys, datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S, D = os.path.join(ROOT, "gpurun_out", "r06"), os.path.join(ROOT, "profiles")
date = sys.argv[1] if len(sys.argv) > 1 else datetime.date.today().isoformat()
def rd(n): return open(os.path.join(S, n)).read()
def wr(n, txt): open(os.path.join(D, n), "w").write(txt)
for w in ("train_c4", "train_c3", "train_c2", "infer_c5", "train_c4_mono"):      # (absent in the first of the script's two collections)
    if os.path.isfile(os.path.join(S, f"bench_line_{w}.json")) and os.path.getsize(os.path.join(S, f"bench_line_{w}.json")) > 0:
        shutil.copy(os.path.join(S, f"bench_line_{w}.json"), os.path.join(D, f"r06_bench_line_{w}.json"))
for tag, w in (("c4", "train_c4"), ("c3", "train_c3"), ("c2", "train_c2")):
    wr(f"r06_kernel_trace_stats_bench_{tag}.txt", f"rocprofv3 --kernel-trace of: python bench.py --workload {w} --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer (round 6, {date})\n" + rd(f"kernel_trace_stats_{tag}.txt"))
    shutil.copy(os.path.join(S, f"in_step_kernel_us_{tag}.json"), os.path.join(D, f"r06_in_step_kernel_us_{tag}.json"))
shutil.copy(os.path.join(D, "r06_kernel_trace_stats_bench_c4.txt"), os.path.join(D, "r06_kernel_trace_stats_bench.txt"))
wr("r06_kernel_trace_by_grid_c4.txt", "per (kernel, grid) census of the same C4 trace (scratch/trace_by_grid.py; grid in threads), round 6\n" + rd("by_grid.txt"))
wr("r06_kernel_trace_stats_serialised_streams.txt", "The same step with every HIP stream serialised (BUCTD_TUNING=1 BUCTD_WGRAD_STREAM=0 BUCTD_BRANCH_STREAMS=0): kernel durations ~ solo; scratch/serial_census.sh (round 6)\n" + rd("serial_bench.txt").strip() + "\n" + rd("serial_stats.txt"))
wr("r06_timeline_concurrency.txt", "GPU timeline of one traced train step (scratch/timeline_gaps.py on the same kernel trace), round 6\n" + rd("timeline.txt"))
wr("r06_critical_path.txt", "Last-arrival critical path of one traced train step (scratch/critical_path.py on the same kernel trace), round 6.\n" + rd("critical_path.txt"))
wr("r06_host_enqueue_split.txt", "Host enqueue time of one train step by phase (scratch/cpu_split.py), round 6\n" + rd("host_split.txt"))
hdr = (f"rocprofv3 PMC passes, roofline kernels of the committed build (bf16x6, 48->48 @96x72, N=32; group launches: + 96->96 @48x36), 1x MI355X, round 6 ({date})\n"
       "command per counter group (one pass each, --kernel-trace only): scratch/pmc_run.sh <tag> bf16x6 <fwd|dgrad|wgrad>; 10 launches each (first dropped); "
       "FETCH_SIZE / WRITE_SIZE in KB (gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md)\n")
wr("r06_pmc_conv3x3.txt", hdr + "\n=== forward (with BN-statistics epilogue) ===\n" + rd("pmc_fwd.txt") + "\n=== data gradient ===\n" + rd("pmc_dgrad.txt")
   + "\n=== weight gradient (kernel + slab reduction) ===\n" + rd("pmc_wgrad.txt")
   + "\n=== GROUP launch, forward: 48->48 @96x72 + 96->96 @48x36 with the statistics accumulators ===\n" + rd("pmc_fwd_group.txt")
   + "\n=== GROUP launch, weight gradients of the same two convolutions (kernel + slab reduction) ===\n" + rd("pmc_wgrad_group.txt"))
shutil.copy(os.path.join(S, "pmc_traffic.json"), os.path.join(D, "r06_pmc_traffic.json"))
print("profiles/ refreshed from", S)
