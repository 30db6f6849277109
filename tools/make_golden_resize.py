#!/usr/bin/env python3
"""Golden vectors for the Resize step: real Pillow (the reference's dependency) on seeded images.
Writes tests/golden/resize.npz: for each case the input image, the requested size and PIL's output."""
import os
import sys

import numpy as np
import PIL
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cases  # noqa: E402

out = {'pillow_version': np.array(PIL.__version__)}
rng = np.random.default_rng(20241004)
for name, h, w, oh, ow in cases.RESIZE_CASES:
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if name.endswith('_smooth'):            # low-frequency content (photographs), not only noise
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(127 + 120 * np.sin(xx / 7.0 + c) * np.cos(yy / 5.0)) for c in range(3)], -1).astype(np.uint8)
    out[name + '/in'] = img
    out[name + '/out'] = np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh), Image.BILINEAR))
np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'resize.npz'), **out)
print('wrote', len(cases.RESIZE_CASES), 'cases with Pillow', PIL.__version__)
