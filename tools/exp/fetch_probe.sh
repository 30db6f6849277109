#!/bin/bash
# FETCH_SIZE of every kernel of tools/exp/blur_probe (run on the GPU box): calibrates the counter on the linear copy
# (known bytes) and compares the blur variants.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_probe -- $R/tools/exp/blur_probe $R/3d-fm-gan_amd/csrc/libfmgan_hip.so > $R/gpurun_out/fetch_probe.txt 2>&1
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/fetch_probe/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="FETCH_SIZE":
            agg[(r["Kernel_Name"][:70], r["Grid_Size"], r.get("LDS_Block_Size",""))].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])/len(kv[1])):
    print(f"{k[0]:70s} grid {k[1]:>9s} lds {k[2]:>6s} n {len(v):3d} FETCH KiB {sum(v)/len(v):12.0f}")
PY
