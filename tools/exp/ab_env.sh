#!/bin/bash
# A/B of one environment switch on bench.py (GPU box): bash tools/exp/ab_env.sh VAR "v1 v2" [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  env $VAR=$v python3 $R/bench.py --no-cpu-baseline --no-secondary --no-train "$@" > $R/gpurun_out/ab_${VAR}_${v}_$rep.json 2> $R/gpurun_out/ab_${VAR}_$v.err
  python3 - <<PY
import json
d=json.loads(open("$R/gpurun_out/ab_${VAR}_${v}_$rep.json").read().strip().splitlines()[-1])
print("$VAR=$v rep $rep:", round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms/step")
PY
done; done
