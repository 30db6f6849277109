#!/usr/bin/env python3
"""Turn the output of tools/profile_mfma.sh (gpurun_out/<log>) into profiles/r01_mfma_counters.md.
usage: python tools/summarize_mfma.py gpurun_out/mfma.log <tag>"""
import ast
import re
import sys

txt = open(sys.argv[1]).read()
tag = sys.argv[2] if len(sys.argv) > 2 else ''
first, second = txt.split('## pmc_mfma2')


def parse(block):
    out = {}
    for line in block.splitlines():
        m = re.match(r"\('void \(anonymous namespace\)::(modconv_mfma_f32<[^>]*>?)[^']*', '(\d+)'\) (\{.*\})", line)
        if m:
            name = m.group(1) if m.group(1).endswith('>') else m.group(1).rstrip(', ') + '>'
            out[(name, m.group(2))] = ast.literal_eval(m.group(3))
    return out


a, b = parse(first), parse(second)
print(f'# MFMA utilisation of the modulated-conv kernels (rocprofv3 --pmc, tools/profile_mfma.sh {tag}, bench_kernels.py conv, B=8)\n')
print('MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs); FLOPs = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512')
print('(equals the algorithmic 2*9*Cin*Cout*B*H*W of the layer: the kernel issues no wasted MFMAs except in thin edge tiles).')
print('Template arguments: <MODE, RM, RNP, WM, WN, RGB epilogue, min blocks per CU, channels per chunk, PIPE (0 register / 1 LDS-DMA)>;')
print('tile = 32*RM*WM channels x 32*RNP*WN positions.  (The mode-1 counts include the MFMAs the thin strips still issue.)\n')
print('| kernel | grid (threads) | MFMA busy cycles | GUI active (sum of 8 XCDs) | MfmaUtil % | GFLOP by counter | wait/wave cycles % |')
print('|---|---|---|---|---|---|---|')
for k, v in a.items():
    util = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024) * 100
    w = b.get(k, {})
    gf = w.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0) * 512 / 1e9
    wait = 100 * w.get('SQ_WAIT_ANY', 0) / max(1, w.get('SQ_WAVE_CYCLES', 1))
    print(f"| {k[0]} | {k[1]} | {v['SQ_VALU_MFMA_BUSY_CYCLES']} | {v['GRBM_GUI_ACTIVE']} | {util:.1f} | {gf:.1f} | {wait:.0f} |")
