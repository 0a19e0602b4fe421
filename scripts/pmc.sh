#!/bin/bash
# scripts/pmc.sh TAG WORKLOAD -- SQ/GRBM counter passes for the FIR kernel (own runs,
# never combined with tracing).  Prints per-launch averages for fir_* kernels.
set -u
TAG=${1:-pmc}; WL=${2:-c2}
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline ${EXTRA:-}"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_WAVES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; exit 1; }
done
python3 scripts/summarize_pmc.py "$OUT"
