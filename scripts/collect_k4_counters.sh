#!/bin/bash
# Run ON THE GPU BOX (through gpurun): what the K4 tile kernel (score_topk_bf16_kernel<refine>) spends its cycles on, at both
# benchmark shapes.  One rocprofv3 pass per counter group (kernel trace only, as the pool requires);
# summarise with `python scripts/summarize_k4_counters.py gpurun_out/k4c_<tag> <tag>` -> profiles/<tag>_pmc_k4.json.
set -u
TAG=${1:-r06}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/k4c_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM"
         "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
         "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH")
for SHAPE in ml nf; do
  if [ $SHAPE = ml ]; then ARGS=""; else ARGS="480189 17770 128 30"; fi
  for C in "${GROUPS_[@]}"; do
    N=$(echo $C | tr ' ' '+')
    REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/${SHAPE}_$N" -o b -- \
        python "$REPO/scripts/probe_topk.py" $ARGS > "$OUT/${SHAPE}_$N.out" 2> "$OUT/${SHAPE}_$N.err" < /dev/null
  done
done
find "$OUT" -type f \( -name "*.db" -o -name "*agent_info*" \) -delete
find "$OUT" -name "*.csv" -size +8M -exec gzip -9 {} \;
du -sh "$OUT"
