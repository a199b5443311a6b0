"""the head of HRNet-W48 (stem + layer1 + transition1) alone, forward + backward, serial streams:
python scratch/time_head.py   (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, torch
os.environ.setdefault("BUCTD_BRANCH_STREAMS", "0"); os.environ.setdefault("BUCTD_WGRAD_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
dev = torch.device("cuda:0")
cfg = bench.coam_w48_cfg(32)
net = models.pose_hrnet_coam.get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net); model.flatten()
x = torch.randn(32, 384, 288, 3, device=dev)
def run():
    y = net.stem(x)
    ys = net.enter_stage(2, y, first=True)
    g = [torch.ones_like(t) for t in ys]
    torch.autograd.backward(ys, g)
    ops.wait_side_stream()
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"head fwd + bwd: {e0.elapsed_time(e1) / 10:.2f} ms")
