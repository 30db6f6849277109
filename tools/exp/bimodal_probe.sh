#!/bin/bash
# GPU box: bash tools/exp/bimodal_probe.sh  -> gpurun_out/r03_bimodal.jsonl
# 4 launches back to back, then 4 with a 25 s pause before each (is the alternation a lazily returned previous process?).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_bimodal.jsonl
: > $OUT
for i in 1 2 3 4; do python3 $R/tools/exp/bimodal_probe.py b2b_$i >> $OUT 2>> $R/gpurun_out/r03_bimodal.err || exit 1; done
for i in 1 2 3 4; do sleep 25; python3 $R/tools/exp/bimodal_probe.py pause_$i >> $OUT 2>> $R/gpurun_out/r03_bimodal.err || exit 1; done
cat $OUT
