#!/bin/bash
# FMGAN_MC_DEBUG exists only in the experiments build: make -C 3d-fm-gan_amd/csrc experiments
export FMGAN_LIB=${FMGAN_LIB:-${GRAFT_REPO_ROOT:-/root/repo}/tools/exp/lib/libfmgan_hip_exp.so}
# PMC breakdown of the MFMA loop (run ON the GPU box): where do the wave-cycles of modconv_mfma_f32 go, next to the
# ideal "ds_read -> MFMA" loop of tools/exp/mfma_lds_probe.  usage: tools/profile_loop.sh <tag>
set -e
TAG=$1
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/loop_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
PB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
PC="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"
run() {  # name, command...
  local name=$1; shift
  for pass in A B C; do
    local ctr; eval ctr=\$P$pass
    rocprofv3 --pmc $ctr --output-format csv -d $OUT/${name}_$pass -- "$@" > $OUT/${name}_$pass.log 2>&1 || true
  done
  echo "$name done"
}
run probe $REPO/tools/exp/mfma_lds_probe
run l64_full python3 $REPO/tools/run_layer.py 64 512 512 0
FMGAN_MC_DEBUG=3 run l64_nostage python3 $REPO/tools/run_layer.py 64 512 512 0
run t128_full python3 $REPO/tools/run_layer.py 64 512 256 1
FMGAN_MC_DEBUG=3 run t128_nostage python3 $REPO/tools/run_layer.py 64 512 256 1
python3 - <<PY
import csv, glob, collections, os
out = '$OUT'
for name in ('probe', 'l64_full', 'l64_nostage', 't128_full', 't128_nostage'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{out}/{name}_*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'modconv_mfma' in k or k.startswith('void P<') or ' P<' in k:
                agg[(k[:70], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print('##', name)
    for k, v in sorted(agg.items()):
        c = {n: sum(x) / len(x) for n, x in v.items()}
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        print(k)
        print('   ', {n: round(x) for n, x in sorted(c.items())})
        print(f"    wait_any/wave {c.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst/wave {c.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}  "
              f"active/wave {c.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  mfma_busy/(gui/8*1024) "
              f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(1.0, c.get('GRBM_GUI_ACTIVE', 0) / 8 * 1024):.3f}  "
              f"lds_conflict/lds_active {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0)):.3f}")
PY
