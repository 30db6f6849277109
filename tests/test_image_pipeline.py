"""The steps either side of the path: Resize + ToTensor + Normalize on the way in, pair swapping.

Resize is Pillow's fixed-point two-pass bilinear resampler (byte arithmetic: the bar is bit equality).  The oracle
(oracle/fmgan_oracle.c::oracle_resize_bilinear_u8) is pinned by tests/golden/resize.npz — outputs of the real Pillow
written by tools/make_golden_resize.py — and, when PIL is importable, by a live comparison.
"""
import numpy as np
import pytest
import torch

import cases


def dev():
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def rz_golden():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resize.npz'))


@pytest.mark.parametrize('c', cases.RESIZE_CASES, ids=lambda c: c[0])
def test_resize_oracle_matches_pillow_golden(c, rz_golden):
    from oracle import c_oracle
    name, h, w, oh, ow = c
    got = c_oracle.resize_bilinear_u8(rz_golden[name + '/in'], oh, ow)
    np.testing.assert_array_equal(got, rz_golden[name + '/out'])


def test_resize_oracle_matches_live_pillow():
    Image = pytest.importorskip('PIL.Image')
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    for h, w, oh, ow in [(256, 256, 64, 64), (61, 97, 200, 13), (300, 200, 128, 85), (8, 8, 8, 8), (40, 30, 40, 31)]:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh), Image.BILINEAR))
        np.testing.assert_array_equal(c_oracle.resize_bilinear_u8(img, oh, ow), ref)


def test_resize_random_shapes_oracle_and_plan_match_live_pillow():
    """Seeded sweep over 60 random size pairs (shrinking, enlarging, mixed, degenerate 1-pixel axes): the C oracle and
    the product's host plan (applied in numpy) both reproduce the installed Pillow byte for byte."""
    Image = pytest.importorskip('PIL.Image')
    from oracle import c_oracle
    from op import _native
    rng = np.random.default_rng(77)
    for _ in range(60):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        oh, ow = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img, 'RGB').resize((ow, oh), Image.BILINEAR))
        np.testing.assert_array_equal(c_oracle.resize_bilinear_u8(img, oh, ow), ref, err_msg=f'oracle {h}x{w}->{oh}x{ow}')
        np.testing.assert_array_equal(_numpy_from_plan(img, _native.resize_plan(h, w, oh, ow), oh, ow), ref,
                                      err_msg=f'plan {h}x{w}->{oh}x{ow}')


def _numpy_from_plan(img, plan, oh, ow):
    """Apply a plan produced by the PRODUCT's host code with plain numpy integer arithmetic."""
    p = plan.numpy().astype(np.int64)
    ksx, ksy = int(p[0]), int(p[1])
    bx = p[8:8 + 2 * ow].reshape(ow, 2)
    kx = p[8 + 2 * ow:8 + 2 * ow + ow * ksx].reshape(ow, ksx)
    o = 8 + 2 * ow + ow * ksx
    by = p[o:o + 2 * oh].reshape(oh, 2)
    ky = p[o + 2 * oh:o + 2 * oh + oh * ksy].reshape(oh, ksy)
    h, w, c = img.shape
    tmp = np.zeros((h, ow, c), np.int64)
    for x in range(ow):
        a, n = bx[x]
        tmp[:, x] = np.clip(((1 << 21) + (img[:, a:a + n].astype(np.int64) * kx[x, :n, None]).sum(1)) >> 22, 0, 255)
    out = np.zeros((oh, ow, c), np.int64)
    for y in range(oh):
        a, n = by[y]
        out[y] = np.clip(((1 << 21) + (tmp[a:a + n] * ky[y, :n, None, None]).sum(0)) >> 22, 0, 255)
    return out.astype(np.uint8)


@pytest.mark.parametrize('c', cases.RESIZE_CASES, ids=lambda c: c[0])
def test_resize_plan_is_host_logic_and_reproduces_pillow(c, rz_golden):
    """fmgan_resize_plan runs without a GPU; its tables, applied in numpy, give Pillow's bytes."""
    from op import _native
    name, h, w, oh, ow = c
    plan = _native.resize_plan(h, w, oh, ow)
    assert plan.dtype == torch.int32 and tuple(plan[2:6].tolist()) == (h, w, oh, ow)
    np.testing.assert_array_equal(_numpy_from_plan(rz_golden[name + '/in'], plan, oh, ow), rz_golden[name + '/out'])
    # every row of weights sums to 1.0 in 22-bit fixed point (within the rounding of its taps)
    ksx = int(plan[0])
    kx = plan[8 + 2 * ow:8 + 2 * ow + ow * ksx].reshape(ow, ksx).sum(1)
    assert int((kx - (1 << 22)).abs().max()) <= ksx


def test_resize_output_size_rule():
    from op import _native
    from oracle import c_oracle
    for h, w, size in [(1024, 1024, 256), (256, 256, 256), (300, 200, 128), (200, 300, 128), (101, 77, 50), (5, 1000, 3),
                       (77, 77, 78)]:
        assert _native.resize_output_size(h, w, size) == c_oracle.resized_output_size(h, w, size)
    assert _native.resize_output_size(300, 200, 128) == (192, 128)
    with pytest.raises(RuntimeError):
        _native.resize_output_size(0, 5, 3)


def test_resize_requires_gpu_and_checks_arguments():
    from op import _native
    with pytest.raises(RuntimeError):
        _native.resize_images(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), 4, 4)        # CPU tensor: no CPU path
    st = _native.lib().fmgan_resize_bilinear_u8(None, None, None, None, 1, 8, 8, 4, 4, 0.5, 0.5, None)
    assert st == -1
    assert _native.lib().fmgan_resize_plan_ints(0, 8, 4, 4) == 0


def test_pair_swapping_matches_reference_semantics():
    import dataset
    assert dataset.Swap_List_Pair(range(6)) == [1, 0, 3, 2, 5, 4]
    with pytest.raises(IndexError):
        dataset.Swap_List_Pair(range(3))
    photo = torch.arange(4 * 3).reshape(4, 3).float()
    render = photo + 100
    it = lambda: iter([(photo.clone(), render.clone())])
    g, r, ref = dataset.Data_Loading(it(), None, False, 'cpu')
    assert torch.equal(g, photo) and torch.equal(r, render) and torch.equal(ref, photo)
    g, r, ref = dataset.Data_Loading(None, it(), True, 'cpu')
    assert torch.equal(g, photo) and torch.equal(r, render[[1, 0, 3, 2]]) and torch.equal(ref, photo[[1, 0, 3, 2]])
    g, r, ref = dataset.Data_Loading(None, None, True, 'cpu', extreme_loader=it(), extreme_ds_flag=True)
    assert torch.equal(g, photo[[0, 2]]) and torch.equal(r, render[[1, 3]]) and torch.equal(ref, photo[[1, 3]])


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('c', cases.RESIZE_CASES, ids=lambda c: c[0])
def test_hip_resize_golden(c, rz_golden):
    from op import _native
    name, h, w, oh, ow = c
    img = torch.from_numpy(rz_golden[name + '/in']).to(dev())
    batch = torch.stack([img, img.flip(0), img.flip(1)])           # three different images of the same size
    got = _native.resize_images(batch, oh, ow).cpu().numpy()
    np.testing.assert_array_equal(got[0], rz_golden[name + '/out'])
    from oracle import c_oracle
    np.testing.assert_array_equal(got[1], c_oracle.resize_bilinear_u8(rz_golden[name + '/in'][::-1], oh, ow))
    np.testing.assert_array_equal(got[2], c_oracle.resize_bilinear_u8(rz_golden[name + '/in'][:, ::-1], oh, ow))
    # fused Resize + ToTensor + Normalize == resize, then the (already pinned) images_to_tensor
    t = _native.resize_images(batch, oh, ow, to_tensor=True)
    assert torch.equal(t, _native.images_to_tensor(torch.from_numpy(got).to(dev())))


@pytest.mark.gpu
def test_hip_resize_ffhq_shape_vs_oracle_and_properties():
    """1024^2 -> 256^2 (FFHQ sources into a 256^2 model): one image against the oracle; then properties at batch
    size: constant images stay constant, identity resize returns the input, the batch is processed image by image."""
    from op import _native
    from oracle import c_oracle
    from Util import image_io
    g = torch.Generator().manual_seed(3)
    imgs = torch.randint(0, 256, (6, 1024, 1024, 3), dtype=torch.uint8, generator=g)
    d = imgs.to(dev())
    out = image_io.resize_images(d, 256)
    assert tuple(out.shape) == (6, 256, 256, 3)
    np.testing.assert_array_equal(out[4].cpu().numpy(), c_oracle.resize_bilinear_u8(imgs[4].numpy(), 256, 256))
    assert torch.equal(image_io.resize_images(d[4:5], 256)[0], out[4])
    const = torch.full((2, 300, 200, 3), 0, dtype=torch.uint8, device=dev())
    const[1] = 201
    r = image_io.resize_images(const, 128)
    assert tuple(r.shape) == (2, 192, 128, 3) and int(r[0].max()) == 0 and int(r[1].min()) == 201 and int(r[1].max()) == 201
    assert image_io.resize_images(d, 1024) is d                                        # Resize(size) no-op rule
    same = _native.resize_images(d[:1], 1024, 1024)                                    # explicit identity resample
    assert torch.equal(same, d[:1])
    t = image_io.load_transform(d, 256)
    assert tuple(t.shape) == (6, 3, 256, 256) and torch.equal(t, _native.images_to_tensor(out))
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0


@pytest.mark.gpu
def test_hip_resize_random_shapes_vs_oracle():
    """Seeded sweep: every kernel variant (3-, 5-, 9-tap dword path, byte path for larger shrink factors), ragged tile
    edges, batches of different images — bit equality with the oracle (itself pinned to Pillow)."""
    from op import _native
    from oracle import c_oracle
    rng = np.random.default_rng(123)
    for _ in range(25):
        h, w = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        oh, ow = int(rng.integers(1, 100)), int(rng.integers(1, 100))
        b = int(rng.integers(1, 4))
        imgs = rng.integers(0, 256, (b, h, w, 3), dtype=np.uint8)
        got = _native.resize_images(torch.from_numpy(imgs).to(dev()), oh, ow).cpu().numpy()
        for i in range(b):
            np.testing.assert_array_equal(got[i], c_oracle.resize_bilinear_u8(imgs[i], oh, ow),
                                          err_msg=f'{h}x{w}->{oh}x{ow} image {i}')
