"""does the first GPU process on a fresh box run slower, and does it recover inside the process?  python scratch/first_run_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_import = time.time()
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
dev = torch.device("cuda:0")
WL = os.environ.get("WL", "train_c4")
cfg = bench.TRAIN_WORKLOADS[WL][0](32)
net = getattr(models, bench.TRAIN_WORKLOADS[WL][1]).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net); opt = engine.get_optimizer(cfg, model); model.flatten()
x, tgt, wt = bench.synthetic_batch(cfg, 32, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward(); opt.step()
import contextlib
mp = os.environ.get("MAINPRIO")
ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(mp))) if mp is not None else contextlib.nullcontext()
ctx.__enter__()
for blk in range(int(os.environ.get("BLOCKS", "12"))):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    clk = os.popen("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | head -2 | tr -s ' ' | cut -d: -f2- | tr '\n' ' '").read().strip()
    ms = torch.cuda.memory_stats()
    print(f"block {blk}: {dt * 1e3:7.2f} ms/step  ({32 / dt:6.1f} img/s)  reserved {ms['reserved_bytes.all.current'] / 2**30:6.2f} GB  live {ms['allocated_bytes.all.current'] / 2**30:6.2f} (peak {ms['allocated_bytes.all.peak'] / 2**30:6.2f})  inactive-split {ms['inactive_split_bytes.all.current'] / 2**30:6.2f}  segments {ms['segment.all.current']}  device mallocs {ms.get('num_device_alloc', -1)} frees {ms.get('num_device_free', -1)}  {clk}", flush=True)
