"""cProfile of the host side of five eager steps (run on the GPU box): where the 12-16 ms of Python/launch time go."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
nets = bench.build_models(1024, d)
step, _ = bench.make_step(nets, 8, d, 0)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
