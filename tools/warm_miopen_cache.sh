#!/bin/bash
# Run ON an MI355X box (via gpurun): fill a MIOpen kernel cache for every convolution bench.py and the GPU tests use,
# under gpurun_out/miopen_cache (copied back by gpurun; move it to 3d-fm-gan_amd/miopen_cache, git-ignored).
# MIOpen in this image has no pre-built gfx950 kernels: without the cache each fresh box spends minutes in hipRTC.
set -e
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/miopen_cache
mkdir -p $OUT/db $OUT/cache
export MIOPEN_USER_DB_PATH=$OUT/db MIOPEN_CUSTOM_CACHE_DIR=$OUT/cache
cd $REPO
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/warm_bench.json 2> gpurun_out/warm_bench.err || true
tail -3 gpurun_out/warm_bench.err
du -sh $OUT $OUT/db $OUT/cache
ls -la $OUT/db $OUT/cache | head -20
