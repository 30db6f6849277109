#!/bin/bash
# MFMA-utilisation counters for the modulated-conv kernels (run ON the GPU box). usage: tools/profile_mfma.sh <tag>
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/tools/bench_kernels.py conv > $OUT/mfma_bench.txt 2> $OUT/pmc_mfma.err || true
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_mfma2 -- python3 $REPO/tools/bench_kernels.py conv > $OUT/mfma_bench2.txt 2> $OUT/pmc_mfma2.err || true
python3 - <<PY
import csv, glob, collections
for tag in ('pmc_mfma', 'pmc_mfma2'):
    fs = glob.glob('$OUT/' + tag + '/**/*counter_collection.csv', recursive=True)
    if not fs:
        print(tag, 'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        if 'modconv_mfma' in r['Kernel_Name']:
            agg[(r["Kernel_Name"][:75], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('##', tag)
    for k, v in sorted(agg.items(), key=lambda kv: -max(sum(x) for x in kv[1].values()))[:14]:
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
grep -i "mfma" $OUT/counters.txt | head -20
