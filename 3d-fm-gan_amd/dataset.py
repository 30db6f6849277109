"""Batch assembly either side of the path (reference: dataset.py:343-406).

The reference swaps the members of each (a, b) identity pair with a Python list + a numpy round trip through host
memory (`g_input.detach().cpu().numpy()[swap_list]`).  Here the swap is one index_select on whatever device the batch
already lives on, so a loader that ships uint8 images to the GPU (Util/image_io.load_transform) never comes back.
"""
import torch


def Swap_List_Pair(idx_list):
    """[0, 1, 2, 3, ...] -> [1, 0, 3, 2, ...] (dataset.py:343-358); the length must be even."""
    idx_list = list(idx_list)
    if len(idx_list) % 2:
        raise IndexError('Swap_List_Pair: odd number of items')
    return [idx_list[i ^ 1] for i in range(len(idx_list))]


def _swap_pairs(t):
    n = t.shape[0]
    if n % 2:
        raise IndexError('pair swap: odd batch')
    idx = torch.arange(n, device=t.device) ^ 1
    return t.index_select(0, idx)


def Data_Loading(rec_loader, ds_loader, ds_flag, device, extreme_loader=None, extreme_ds_flag=False):
    """One training batch (dataset.py:361-406, the `ds_dataset_type is None` branch):
    reconstruction   -> (photo, render, target = photo)
    dual supervision -> (photo, render of the PARTNER image, target = partner photo)
    extreme pose     -> the same, keeping only the even members of each pair.
    Loaders yield (photo, render) batches; tensors are moved to `device` first, then permuted there."""
    if not ds_flag:
        g_input, r_input = next(rec_loader)
        g_input, r_input = g_input.to(device), r_input.to(device)
        return g_input, r_input, g_input.clone()
    loader = extreme_loader if extreme_ds_flag else ds_loader
    g_input, r_input = next(loader)
    g_input, r_input = g_input.to(device), r_input.to(device)
    r_input = _swap_pairs(r_input)
    g_ref = _swap_pairs(g_input)
    if extreme_ds_flag:
        g_input, r_input, g_ref = g_input[0::2], r_input[0::2], g_ref[0::2]
    return g_input, r_input, g_ref
