"""DEBUG: cycle stamps of the persistent 3x3 kernel (wave 0 of workgroups 0..63): python scratch/c3p_stamps.py [set]"""
import os, sys, torch
dev = torch.device("cuda:0")
buf = torch.zeros(64 * 64, dtype=torch.int64, device=dev)
os.environ["C3DBGBUF"] = hex(buf.data_ptr())
sys.argv = [sys.argv[0], "w48", "32"] + (sys.argv[1:] or ["0000"])
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_c3.py"), run_name="__main__")
torch.cuda.synchronize()
b = buf.cpu().view(64, 64)
for s in range(0, 64, 1):
    hw = int(b[s, 63])
    st = [int(x) for x in b[s, :63] if x != 0]
    if len(st) < 2:
        continue
    d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
    print(f"wg {s:2d} cu {(hw >> 8) & 15} se {(hw >> 13) & 7} sh {(hw >> 12) & 1} tg {(hw >> 16) & 15} wave {hw & 15} simd {(hw >> 4) & 3}: total {st[-1] - st[0]}: " + " ".join(str(x) for x in d[:40]))
