#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace --stats, and --pmc FETCH_SIZE / WRITE_SIZE passes) into a small
markdown summary that can be committed under profiles/."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict



def headline(name):
    """The headline blur's kernels: path 1b (aligned-row input: every in-step launch) and path 1 (bench.py's standalone
    op on a contiguous tensor)."""
    return 'ufd_rowmarch_f32<4' in name or 'ufd_dmaring_f32' in name

def find(d, pat):
    return sorted(glob.glob(os.path.join(d, '**', pat), recursive=True))


def short(name, n=110):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    return name if len(name) <= n else name[:n - 3] + '...'


def main(out):
    print(f'# rocprofv3 summary: {os.path.basename(out)}\n')
    for f in find(os.path.join(out, 'trace'), '*kernel_stats.csv'):
        rows = list(csv.DictReader(open(f)))
        # MIOpen's exhaustive find (torch.backends.cudnn.benchmark) times every applicable solver once per process,
        # including its naive reference convs; those launches happen before the timed region and are listed apart
        find_rows = [r for r in rows if r['Name'].startswith('naive_conv')]
        rows = [r for r in rows if not r['Name'].startswith('naive_conv')]
        print('## kernel-trace --stats (whole run: warm-up + MIOpen find + timed steps)\n')
        if find_rows:
            ms = sum(float(r['TotalDurationNs']) for r in find_rows) / 1e6
            print(f'(excluded: {sum(int(r["Calls"]) for r in find_rows)} launches / {ms:.0f} ms of MIOpen naive_conv_* '
                  f'kernels run by the one-off solver search during warm-up; percentages below are of the whole run)\n')
        print('| kernel | calls | total ms | avg us | % |')
        print('|---|---|---|---|---|')
        for r in rows[:40]:
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                  f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
        print()
    # per-dispatch trace: average duration per (kernel, grid) for our kernels
    for f in find(os.path.join(out, 'trace'), '*kernel_trace.csv'):
        agg = defaultdict(list)
        last_headline = None
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp'])):
            nm = r['Kernel_Name']
            if 'ufd_' in nm or 'modconv' in nm or 'fba_' in nm or 'torgb' in nm or 'noise_bias' in nm:
                label = short(nm, 90)
                if headline(nm) and r.get('Grid_Size_X') == '1048576':
                    # the headline blur: launches inside a step (fused epilogue) vs bench.py's back-to-back plain op
                    t = int(r['Start_Timestamp'])
                    in_step = last_headline is None or t - last_headline > 3_000_000
                    last_headline = t
                    label = ('[in a step, fused epilogue] ' if in_step else '[standalone op, back to back] ') + short(nm, 60)
                key = (label, r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('LDS_Block_Size', ''),
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
        marks = [i for i, r in enumerate(rows) if headline(r['Kernel_Name'])]
        big = max((int(rows[i]['Grid_Size_X']) for i in marks), default=0)
        marks = [i for i in marks if int(rows[i]['Grid_Size_X']) == big]
        # bench.py also launches the headline blur back to back (standalone roofline): a step is a pair of marks
        # more than 5 ms apart
        pairs = [(a, b) for a, b in zip(marks, marks[1:])
                 if int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']) > 5_000_000]
        if len(pairs) >= 2:
            step = rows[pairs[-2][0]:pairs[-2][1]]
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


def headline_traffic(out, tag):
    """HBM-side bytes of the headline blur per launch from the two PMC passes, split into the launches inside a step
    (fused noise/bias/lrelu store, strided input) and bench.py's standalone back-to-back launches of the plain op."""
    import json
    res = {}
    for sub, ctr in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
        for f in find(os.path.join(out, sub), '*counter_collection.csv'):
            vals = sorted((int(r['Dispatch_Id']), float(r['Counter_Value'])) for r in csv.DictReader(open(f))
                          if r['Counter_Name'] == ctr and headline(r['Kernel_Name']) and r['Grid_Size'] == '1048576')
            insitu = [v for i, (d, v) in enumerate(vals)
                      if (i == 0 or d - vals[i - 1][0] > 50) and (i + 1 == len(vals) or vals[i + 1][0] - d > 50)]
            alone = [v for d, v in vals if v not in insitu]
            res[ctr] = (insitu, alone)
    if len(res) < 2:
        return
    def mb(fetch, write):
        return (2.0 * sum(fetch) / len(fetch) + sum(write) / len(write)) * 1024.0
    fi, fa = res['FETCH_SIZE']
    wi, wa = res['WRITE_SIZE']
    doc = {
        'kernel': 'ufd_dmaring_f32<true> in a step / ufd_rowmarch_f32<4,true> standalone, [256,1025,1025]->[256,1024,1024] (grid 1048576 threads)',
        'correction': 'gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM) -> x2; WRITE_SIZE exact',
        'in_step': {'what': 'blur + fused noise/bias/lrelu store, aligned-row strided input (as launched by StyledConv)',
                    'FETCH_SIZE_KiB_per_launch': sum(fi) / len(fi), 'WRITE_SIZE_KiB_per_launch': sum(wi) / len(wi),
                    'launches_sampled': [len(fi), len(wi)], 'hbm_bytes_per_launch': mb(fi, wi)},
        'source': f'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, tools/profile_gpu.sh {tag}',
    }
    if fa and wa:
        doc['standalone_op'] = {'what': 'op.upfirdn2d, contiguous input, no epilogue (bench.py roofline_upfirdn2d_op)',
                                'FETCH_SIZE_KiB_per_launch': sum(fa) / len(fa), 'WRITE_SIZE_KiB_per_launch': sum(wa) / len(wa),
                                'launches_sampled': [len(fa), len(wa)], 'hbm_bytes_per_launch': mb(fa, wa)}
    doc['hbm_bytes_per_launch'] = doc['in_step']['hbm_bytes_per_launch']
    # staleness guard read by bench.py::committed_traffic: the number describes THIS version of the kernel source
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = '3d-fm-gan_amd/csrc/upfirdn2d.hip'
    doc['kernel_source'] = src
    doc['kernel_source_sha256'] = hashlib.sha256(open(os.path.join(root, src), 'rb').read()).hexdigest()
    with open(os.path.join(out, 'headline_traffic.json'), 'w') as fh:
        json.dump(doc, fh, indent=1)


if __name__ == '__main__':
    main(sys.argv[1])
    headline_traffic(sys.argv[1], os.path.basename(sys.argv[1].rstrip('/')).replace('prof_', ''))
