"""whole-step hipGraph capture of a train step, single-stream mode: python scratch/graph_c2.py [workload] [fwdbwd|full]"""
import os, sys, time, torch
os.environ.setdefault("BUCTD_BRANCH_STREAMS", "0"); os.environ.setdefault("BUCTD_WGRAD_STREAM", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from buctd_amd import engine, models, ops
from buctd_amd.core.loss import JointsMSELoss
WL = sys.argv[1] if len(sys.argv) > 1 else "train_c2"
mode = sys.argv[2] if len(sys.argv) > 2 else "fwdbwd"
dev = torch.device("cuda:0")
B = int(os.environ.get("B", "32"))
mk, module = bench.TRAIN_WORKLOADS[WL][0], bench.TRAIN_WORKLOADS[WL][1]
cfg = mk(B)
net = getattr(models, module).get_pose_net(cfg, is_train=True).to(dev).train()
model = engine.DataParallel(net)
opt = engine.get_optimizer(cfg, model)
x, tgt, wt = bench.synthetic_batch(cfg, B, dev, 1)
crit = JointsMSELoss(True)
def step():
    loss = crit(model(x), tgt, wt); opt.zero_grad(); loss.backward()
    if mode == "full": opt.step()
    return loss
for _ in range(3): l = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): l = step()
torch.cuda.synchronize(); print(f"eager (single stream): {(time.time() - t0) / 10 * 1e3:.2f} ms/step, loss {l.item():.6f}", flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
print("begin capture", flush=True)
with torch.cuda.graph(g):
    lg = step()
print("captured", flush=True)
for _ in range(3): g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10): g.replay()
torch.cuda.synchronize(); print(f"graph replay: {(time.time() - t0) / 10 * 1e3:.2f} ms/step, loss {lg.item():.6f}", flush=True)
