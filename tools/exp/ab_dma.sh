#!/bin/bash
# NOTE (round 3): the FMGAN_UFD_DMA environment switch this script drives was removed from the library; it documents how
# the round-2 numbers in profiles/r02_blur_probe.md were taken (tree ab53d56).  A/B now: force_path 4 / 5 through the ABI.
# In-step A/B of the headline blur's launch modes (FMGAN_UFD_DMA): bench.py's roofline object + FETCH_SIZE per launch.
# usage (GPU box): bash tools/exp/ab_dma.sh "1 0" [reps]
# NOTE: the in-step timing of this kernel is BIMODAL between consecutive process launches (4.65 vs 5.1 TB/s on one box,
# alternating from one launch of the same command to the next: physical placement of the two 1 GB buffers), so the
# modes are run in a palindromic order (a b b a a b b a ...) and every variant needs several launches.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODES=${1:-"1 0"}
REPS=${2:-2}
rm -f $R/gpurun_out/ab_dma_*.json
FWD="$MODES"; REV=$(echo $MODES | tr ' ' '\n' | tac | tr '\n' ' ')
i=0
for rep in $(seq 1 $REPS); do for seq in "$FWD" "$REV"; do for m in $seq; do
  i=$((i+1))
  FMGAN_UFD_DMA=$m python3 $R/bench.py --no-cpu-baseline --no-secondary --no-train > $R/gpurun_out/ab_dma_${m}_$i.json 2> $R/gpurun_out/ab_dma_$m.err
done; done; done
python3 - <<PY
import json,glob,re
R="$R"
res={}
for f in sorted(glob.glob(f"{R}/gpurun_out/ab_dma_*_*.json"), key=lambda f:int(re.findall(r'_(\d+)\.json',f)[0])):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    m=f.split("/")[-1].split("_")[2]
    res.setdefault(m,[]).append(round(d["roofline"]["achieved"]))
    print(f.split("/")[-1],"pairs/s",round(d["value"],1),"headline GB/s",round(d["roofline"]["achieved"]))
for m,v in res.items(): print("mode",m,"launches",v,"mean",round(sum(v)/len(v)))
PY
