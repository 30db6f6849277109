#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel-trace stats of any bench.py workload (forward legs: no child processes).
# usage: tools/profile_wl.sh <tag> <workload> [bench args]
set -e
TAG=$1; WL=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $REPO/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-train "$@" > $OUT/warm.json 2> $OUT/warm.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-train "$@" > $OUT/bench.json 2> $OUT/trace.err
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats: bench.py --workload $WL --steps 5 --warmup 2  (sum of kernel time {tot/1e6:.1f} ms)")
    print("| kernel | calls | total ms | avg us | % |"); print("|---|---|---|---|---|")
    for r in rows[:40]:
        print(f"| {r['Name'][:140]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
PY
tail -1 $OUT/bench.json | cut -c1-300
