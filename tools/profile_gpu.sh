#!/bin/bash
# Run ON the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# usage: tools/profile_gpu.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-secondary $@"
# un-profiled run first: MIOpen's exhaustive find (cudnn.benchmark) writes its user find-db, so the profiled runs
# below replay the chosen solvers instead of filling the trace with naive/candidate conv kernels
python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary $@ > $OUT/bench_warm.json 2> $OUT/warm.err || true
echo "warm done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $@ > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $@ > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
echo "write done"
python3 $REPO/tools/summarize_prof.py $OUT > $OUT/summary.md
tail -60 $OUT/summary.md
