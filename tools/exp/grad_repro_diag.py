#!/usr/bin/env python3
"""Which Generator gradients are not bit-identical run to run, and by how much?  (GPU box)
    python tools/exp/grad_repro_diag.py [runs] [hip_wgrad]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import test_hip_train as H  # noqa: E402
from op import modconv  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
if len(sys.argv) > 2:
    modconv.HIP_WGRAD = int(sys.argv[2])
print(f'HIP_WGRAD = {modconv.HIP_WGRAD}, {runs} runs')
poison = os.environ.get('POISON', '0') != '0'
print('poisoned allocator cache (every cached block NaN-filled before each run):', poison)
ref = None
for r in range(runs):
    if poison:
        # hand NaN-filled blocks of many sizes back to the caching allocator: a kernel that reads a torch.empty() buffer it
        # did not fully write now produces NaN instead of whatever the previous owner left there
        torch.cuda.empty_cache()
        junk = [torch.full((n,), float('nan'), device='cuda') for n in
                [1 << k for k in range(10, 29)] + [3 << k for k in range(10, 27)] + [(1 << 28) + 12345] * 4]
        del junk
    G, img, _, _ = H.run_generator_only()
    g = {n: p.grad.clone() for n, p in G.named_parameters() if p.grad is not None}
    bad = [n for n, t in g.items() if not torch.isfinite(t).all()]
    if bad or not torch.isfinite(img).all():
        print(f'run {r}: NON-FINITE values in', bad[:10], 'image finite:', bool(torch.isfinite(img).all()))
    del G
    if ref is None:
        ref, ref_img = g, img.clone()
        continue
    diffs = []
    for n in g:
        if not torch.equal(g[n], ref[n]):
            d = (g[n] - ref[n]).abs()
            diffs.append((n, int((d > 0).sum()), g[n].numel(), float(d.max() / ref[n].abs().max())))
    print(f'run {r}: image equal {torch.equal(img, ref_img)}; {len(diffs)} tensors differ from run 0')
    for n, k, tot, rel in diffs:
        print(f'    {n}: {k} of {tot} elements, max diff {rel:.2e} of max')
