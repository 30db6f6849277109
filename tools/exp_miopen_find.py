import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import resnet_encoder
from psp_encoder_model.encoders import psp_encoders
def timeit(fn, iters=10, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = (os.environ.get('BENCHMARK', '0') == '1')
torch.manual_seed(0)
x = torch.rand(B, 3, 256, 256, device=d) * 2 - 1
with torch.no_grad():
    m = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=18)).to(d).eval()
    t0 = time.perf_counter(); m(x); torch.cuda.synchronize(); first = time.perf_counter() - t0
    print(f'benchmark={torch.backends.cudnn.benchmark} psp18 B={B}: first call {first:.1f} s, steady {timeit(lambda: m(x)):.2f} ms')
    r = resnet_encoder.resnet18(tensor_encoding=True).to(d).eval()
    print(f'  resnet18: {timeit(lambda: r(x)):.2f} ms')
