#!/bin/bash
# Run ON the GPU box: rocprofv3 --kernel-trace --stats of an arbitrary python command, top kernels as markdown.
# usage: tools/profile_cmd.sh <tag> <python args...>      e.g.  PHASES=R1 tools/profile_cmd.sh r1_1024 tools/trainstep_phases.py 1024
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS=()
for a in "$@"; do case "$a" in /*) ARGS+=("$a");; *.py) ARGS+=("$REPO/$a");; *) ARGS+=("$a");; esac; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "${ARGS[@]}" > $OUT/out.json 2> $OUT/trace.err
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats: python3 $*  (sum of kernel time {tot/1e6:.1f} ms)")
    print("| kernel | calls | total ms | avg us | % |"); print("|---|---|---|---|---|")
    for r in rows[:40]:
        print(f"| {r['Name'][:160]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
PY
tail -1 $OUT/out.json | cut -c1-600
