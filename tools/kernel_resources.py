#!/usr/bin/env python3
"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (VGPR / AGPR / spills / occupancy per kernel).
    hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip 2> res.txt ; python tools/kernel_resources.py res.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0].strip()
    try:
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip() or name
    except Exception:
        pass
    name = name.replace('(anonymous namespace)::', '')
    if flt not in name:
        continue

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print(f"{name[:100]:100s} VGPR {g('    VGPRs'):>3} AGPR {g('AGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize .bytes/lane.'):>4} "
          f"occ {g('Occupancy .waves/SIMD.')}")
