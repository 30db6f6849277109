#!/bin/bash
# In-step A/B of the headline blur's launch modes (FMGAN_UFD_DMA): bench.py's roofline object + FETCH_SIZE per launch.
# usage (GPU box): bash tools/exp/ab_dma.sh "1 n x 0"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODES=${1:-"1 n x 0"}
rm -f $R/gpurun_out/ab_dma_*.json
for rep in 1 2; do for m in $MODES; do
  FMGAN_UFD_DMA=$m python3 $R/bench.py --no-cpu-baseline --no-secondary --no-train > $R/gpurun_out/ab_dma_${m}_$rep.json 2> $R/gpurun_out/ab_dma_$m.err
done; done
for m in $MODES; do
  export FMGAN_UFD_DMA=$m
  rm -rf $R/gpurun_out/ab_fetch_$m
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ab_fetch_$m -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-train > /dev/null 2> $R/gpurun_out/ab_fetch_$m.err
done
python3 - <<PY
import json,glob,csv
R="$R"
for f in sorted(glob.glob(f"{R}/gpurun_out/ab_dma_*_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1],"pairs/s",round(d["value"],1),"headline GB/s",round(d["roofline"]["achieved"]),"frac",round(d["roofline"]["frac"],4))
for d in sorted(glob.glob(f"{R}/gpurun_out/ab_fetch_*/")):
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and ("ufd_dmaring" in r["Kernel_Name"] or "ufd_rowmarch_f32<4" in r["Kernel_Name"]) and r["Grid_Size"]=="1048576"]
        v=v[:3]   # the first launches are the in-step ones (bench.py's standalone op comes last)
        print(d.split("/")[-2],"FETCH_SIZE KiB/launch",[round(x) for x in v])
PY
