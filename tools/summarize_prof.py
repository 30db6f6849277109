#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats, and --pmc FETCH_SIZE / WRITE_SIZE passes) into a small
markdown summary that can be committed under profiles/."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    return sorted(glob.glob(os.path.join(d, '**', pat), recursive=True))


def short(name, n=110):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return name if len(name) <= n else name[:n - 3] + '...'


def main(out):
    print(f'# rocprofv3 summary: {os.path.basename(out)}\n')
    for f in find(os.path.join(out, 'trace'), '*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        print('## kernel-trace --stats (all kernels, whole run)\n')
        print('| kernel | calls | total ms | avg us | % |')
        print('|---|---|---|---|---|')
        for r in rows[:40]:
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                  f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        print()
    # per-dispatch trace: average duration per (kernel, grid) for our kernels
    for f in find(os.path.join(out, 'trace'), '*kernel_trace.csv'):
        agg = defaultdict(list)
        for r in csv.DictReader(open(f)):
            nm = r['Kernel_Name']
            if 'ufd_' in nm or 'modconv' in nm or 'fba_' in nm or 'torgb' in nm or 'noise_bias' in nm:
                key = (short(nm, 90), r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('LDS_Block_Size', ''),
                       r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''))
                agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        print('## our kernels per launch shape (kernel, grid.x, LDS, VGPR, AGPR)\n')
        print('| kernel | grid.x | lds | vgpr | agpr | calls | avg us | total ms |')
        print('|---|---|---|---|---|---|---|---|')
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print(f'| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {len(v)} | {sum(v) / len(v) / 1e3:.1f} | {sum(v) / 1e6:.3f} |')
        print()
    # steady state: kernels between the last two launches of the headline blur = one full step
    for f in find(os.path.join(out, 'trace'), '*kernel_trace.csv'):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
        marks = [i for i, r in enumerate(rows) if 'ufd_rowmarch_f32<4' in r['Kernel_Name']]
        big = max((int(rows[i]['Grid_Size_X']) for i in marks), default=0)
        marks = [i for i in marks if int(rows[i]['Grid_Size_X']) == big]
        if len(marks) >= 3:
            step = rows[marks[-3]:marks[-2]]
            t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
            agg = defaultdict(lambda: [0, 0])
            for r in step:
                k = short(r['Kernel_Name'], 80)
                agg[k][0] += 1
                agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            tot = sum(v[1] for v in agg.values())
            print(f'## one steady-state step (between two headline-blur launches): wall {(t1 - t0) / 1e6:.2f} ms, '
                  f'{len(step)} kernels, sum of kernel time {tot / 1e6:.2f} ms\n')
            print('| kernel | launches | total ms |')
            print('|---|---|---|')
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
                print(f'| {k} | {v[0]} | {v[1] / 1e6:.3f} |')
            print()
    for tag, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        for f in find(os.path.join(out, tag), '*counter_collection.csv'):
            agg = defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] != ctr:
                    continue
                nm = r['Kernel_Name']
                if 'ufd_' in nm or 'fba_' in nm or 'noise_bias' in nm or 'torgb' in nm or 'modconv_mfma' in nm:
                    agg[(short(nm, 90), r.get('Grid_Size', ''))].append(float(r['Counter_Value']))
            print(f'## {ctr} per launch (KiB as reported; gfx950: double FETCH_SIZE for wide coalesced reads)\n')
            print('| kernel | grid | launches | avg value (KiB) | avg MB |')
            print('|---|---|---|---|---|')
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:30]:
                a = sum(v) / len(v)
                print(f'| {k[0]} | {k[1]} | {len(v)} | {a:.0f} | {a * 1024 / 1e6:.1f} |')
            print()


if __name__ == '__main__':
    main(sys.argv[1])
