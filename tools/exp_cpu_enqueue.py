import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
nets = bench.build_models(1024, d)
step, _ = bench.make_step(nets, 8, d, 0)
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('CPU enqueue ms per step:', [round(a * 1e3, 2) for a, b in ts], '| enqueue+GPU ms:', [round(b * 1e3, 2) for a, b in ts])
