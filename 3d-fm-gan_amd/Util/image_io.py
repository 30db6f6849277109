"""Input side of the path on the GPU: uint8 HWC images -> normalised float CHW tensors, i.e. the reference's
transforms.Resize(size) + ToTensor() + Normalize((0.5,)*3, (0.5,)*3) (train_3_encoder.py:233-239) in one pass, so the
loader only ships bytes (4x less PCIe traffic than float tensors, 64x less when 1024^2 sources feed a 256^2 model).
Resize follows PIL.Image.resize(BILINEAR) bit for bit (Pillow's fixed-point two-pass resampler)."""
from op import _native


def images_to_tensor(images_u8_hwc, mean=0.5, std=0.5):
    """ToTensor + Normalize: uint8 [B,H,W,3] -> float32 [B,3,H,W]."""
    return _native.images_to_tensor(images_u8_hwc, mean, std)


def resize_output_size(h, w, size):
    """Output (h, w) of torchvision's Resize(size:int) for an [h, w] image."""
    return _native.resize_output_size(h, w, size)


def resize_images(images_u8_hwc, size):
    """transforms.Resize(size) on a batch of equally sized images: uint8 [B,H,W,3] -> uint8 [B,h',w',3]."""
    b, h, w, _ = images_u8_hwc.shape
    oh, ow = _native.resize_output_size(h, w, size)
    if (oh, ow) == (h, w):
        return images_u8_hwc
    return _native.resize_images(images_u8_hwc, oh, ow)


def load_transform(images_u8_hwc, size, mean=0.5, std=0.5):
    """The reference's whole transform (Resize(size) -> ToTensor -> Normalize) in one kernel:
    uint8 [B,H,W,3] -> float32 [B,3,h',w']."""
    b, h, w, _ = images_u8_hwc.shape
    oh, ow = _native.resize_output_size(h, w, size)
    if (oh, ow) == (h, w):
        return _native.images_to_tensor(images_u8_hwc, mean, std)
    return _native.resize_images(images_u8_hwc, oh, ow, to_tensor=True, mean=mean, std=std)
