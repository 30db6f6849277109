import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch, synth
import stylegan2, resnet_encoder
from psp_encoder_model.encoders import psp_encoders
from Util import streams
from Util.network_util import Forward_Inference_3_Encoder
d = torch.device('cuda', 0)
def load(m, kind, seed):
    m.load_state_dict(synth.state_dict(kind, m.state_dict(), seed=seed)); return m.to(d).eval()
e_tsr = load(resnet_encoder.resnet18(tensor_encoding=True), 'resnet', 5)
e_w = load(resnet_encoder.resnet18(tensor_encoding=False), 'resnet', 6)
e_wp = load(psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=10)), 'psp', 7)
G = load(stylegan2.Generator(64, 512, 2), 'generator', 4)
p = synth.tensor('ovl/photo', (2, 3, 256, 256), dist='uniform').to(d)
r = synth.tensor('ovl/render', (2, 3, 256, 256), dist='uniform').to(d)
lat = synth.tensor('lat', (2, 10, 512)).to(d); tsr = synth.tensor('tsr', (2, 512, 4, 4)).to(d)
def diffs(fn, n=6):
    ref = fn().clone(); out = []
    for _ in range(n):
        out.append(float((fn() - ref).abs().max()))
    return out
with torch.no_grad():
    for en in (True, False):
        streams.ENABLED = en
        print('overlap', en)
        print('  e_tsr', diffs(lambda: e_tsr(p)))
        print('  e_wp ', diffs(lambda: e_wp(p)))
        print('  G    ', diffs(lambda: G(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True, external_input_tensor=tsr, randomize_noise=False)))
        class W(torch.nn.Module):
            def __init__(s, g): super().__init__(); s.module = g
            def forward(s, **kw): return s.module(randomize_noise=False, **kw)
        print('  full ', diffs(lambda: Forward_Inference_3_Encoder(p, r, e_tsr, e_w, e_wp, W(G))))
