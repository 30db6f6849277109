#!/bin/bash
# A/B of one environment switch on bench.py (GPU box): bash tools/exp/ab_env.sh VAR "v1 v2" [bench args]
# Values run in palindromic order (a b b a): some in-step quantities alternate between consecutive process launches
# (profiles/r02_blur_probe.md), which an a-b-a-b order would alias with.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
REV=$(echo $VALS | tr ' ' '\n' | tac | tr '\n' ' ')
i=0
for v in $VALS $REV; do
  i=$((i+1))
  env $VAR=$v python3 $R/bench.py --no-cpu-baseline --no-secondary --no-train "$@" > $R/gpurun_out/ab_${VAR}_${v}_$i.json 2> $R/gpurun_out/ab_${VAR}_$v.err
  python3 - <<PY
import json
d=json.loads(open("$R/gpurun_out/ab_${VAR}_${v}_$i.json").read().strip().splitlines()[-1])
print("$VAR=$v launch $i:", round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms/step")
PY
done
