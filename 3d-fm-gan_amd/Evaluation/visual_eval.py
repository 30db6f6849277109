"""Output side of the path (reference: Evaluation/visual_eval.py:24-38) — only `tensor2im`; the GIF/video drivers of
the reference file are evaluation tooling, out of scope."""
import numpy as np

from op import _native


def tensor2im(image_tensor, imtype=np.uint8, cent=1., factor=255. / 2.):
    """[-1,1] image tensor -> numpy uint8 HWC image of the FIRST sample, as the reference does; the clip/scale/cast
    and the CHW->HWC transposition run in one GPU pass (`tensor2im_batch` converts every sample)."""
    img = _native.tensor_to_images(image_tensor[:1].float(), cent, factor)[0].cpu().numpy()
    return img.astype(imtype)


def tensor2im_batch(image_tensor, cent=1., factor=255. / 2.):
    """uint8 [B,H,W,3] GPU tensor for the whole batch."""
    return _native.tensor_to_images(image_tensor.float(), cent, factor)
