"""Input side of the path on the GPU: uint8 HWC images -> normalised float CHW tensors, i.e. the reference's
transforms.ToTensor() + Normalize((0.5,)*3, (0.5,)*3) (train_3_encoder.py:233-239) in one pass, so the loader only
ships bytes (4x less PCIe traffic than float tensors).  Resize(size) is the identity for the 256^2 datasets the
reference uses (train_3_encoder_hyperparams.py:23) and is not provided."""
from op import _native


def images_to_tensor(images_u8_hwc, mean=0.5, std=0.5):
    return _native.images_to_tensor(images_u8_hwc, mean, std)
