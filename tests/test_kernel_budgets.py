"""Register / spill budgets of the kernels the measured occupancies rely on, read from the code objects inside the built
library (llvm-objdump --offloading + llvm-readelf --notes: no GPU, no recompilation).

DESIGN §3.1 / §3.3 quote blocks per CU per tile (2 / 3 / 4 for the 128- / 64- / 32-channel modulated-conv tiles, 16 waves per
CU for the DMA-ring blur); those follow from `512 / vgpr_count` waves per SIMD.  A compiler or source change that pushes a
kernel over its budget, or makes it spill, silently costs a block per CU — this test makes it loud.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, '3d-fm-gan_amd', 'csrc', 'libfmgan_hip.so')
LLVM = '/opt/rocm/lib/llvm/bin'


@pytest.fixture(scope='module')
def kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, 'llvm-objdump'), os.path.join(LLVM, 'llvm-readelf')
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip('llvm-objdump / llvm-readelf not found')
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    d = tmp_path_factory.mktemp('codeobj')
    shutil.copy(LIB, d / 'lib.so')
    subprocess.run([objdump, '--offloading', str(d / 'lib.so')], check=True, capture_output=True, cwd=d)
    out = {}
    for f in sorted(os.listdir(d)):
        if not f.endswith('gfx950'):
            continue
        notes = subprocess.run([readelf, '--notes', str(d / f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split('- .agpr_count')[1:]:
            name = re.search(r'\.name:\s+(\S+)', blk).group(1)

            def num(key):
                return int(re.search(key + r':\s+(\d+)', blk).group(1))
            out[name] = {'vgpr': num(r'\.vgpr_count'), 'vgpr_spill': num(r'\.vgpr_spill_count'),
                         'scratch': num(r'\.private_segment_fixed_size'), 'lds': num(r'\.group_segment_fixed_size')}
    assert len(out) > 80, f'only {len(out)} kernels found in the library'
    return out


def _one(kernels, needle):
    hits = [k for k in kernels if needle in k]
    assert len(hits) == 1, (needle, hits)
    return kernels[hits[0]]


# (mangled-name fragment, max registers for the quoted blocks per CU = 512 // blocks, spills allowed)
MODCONV = [
    ('modconv_mfma_f32ILi0ELi4ELi2ELi1ELi4ELb0ELi2ELi4ELi1E', 256, False),   # 128 x 256 plain tile: 2 blocks per CU
    ('modconv_mfma_f32ILi0ELi2ELi2ELi1ELi4ELb0ELi3ELi4ELi1E', 170, False),   # 64 x 256 plain tile: 3
    ('modconv_mfma_f32ILi0ELi1ELi2ELi1ELi4ELb0ELi4ELi4ELi1E', 128, False),   # 32 x 256 plain tile, 4-channel chunks: 4
    ('modconv_mfma_f32ILi0ELi1ELi2ELi1ELi4ELb1ELi4ELi4ELi1E', 128, True),    # the same with the ToRGB epilogue (epilogue-only spills)
    ('modconv_mfma_f32ILi1ELi2ELi1ELi1ELi4ELb0ELi2ELi8ELi1E', 256, False),   # transposed 64-channel tile: 2
    ('modconv_mfma_f32ILi1ELi1ELi1ELi1ELi4ELb0ELi4ELi8ELi1E', 128, True),    # transposed 32-channel tile: 4 (7 dwords in the epilogue)
]


@pytest.mark.parametrize('needle,budget,spills_ok', MODCONV, ids=[m[0][16:] for m in MODCONV])
def test_modconv_tiles_fit_their_blocks_per_cu(kernels, needle, budget, spills_ok):
    k = _one(kernels, needle)
    assert k['vgpr'] <= budget, k
    if not spills_ok:
        assert k['vgpr_spill'] == 0 and k['scratch'] == 0, k
    else:
        assert k['vgpr_spill'] <= 32, k


def test_reduced_precision_and_split_operand_kernels_do_not_spill(kernels):
    names = [k for k in kernels if 'modconv_mfma_bf16' in k]
    assert len(names) == 8, names                      # 6 bf16 instantiations + 2 split-operand ones
    for n in names:
        assert kernels[n]['vgpr_spill'] == 0 and kernels[n]['scratch'] == 0 and kernels[n]['vgpr'] <= 256, (n, kernels[n])


def test_streaming_kernels_keep_their_occupancy(kernels):
    # DMA-ring blur with the fused epilogue: 64 registers = 8 waves per SIMD (its ring lives in LDS, not in registers)
    k = _one(kernels, 'ufd_dmaring_f32ILb1ELb0ELb0E')
    assert k['vgpr'] <= 64 and k['vgpr_spill'] == 0 and k['scratch'] == 0, k
    for needle, budget in (('ufd_rowmarch_f32ILi4ELb1E', 128), ('noise_bias_act_f32', 32), ('equal_linear_f32', 32),
                           ('modconv_demod_f32ILb1E', 40), ('fba_f32_inner4ILi30ELb0E', 32)):
        k = _one(kernels, needle)
        assert k['vgpr'] <= budget and k['vgpr_spill'] == 0 and k['scratch'] == 0, (needle, k)
