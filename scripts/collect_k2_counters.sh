#!/bin/bash
# Run ON THE GPU BOX (through gpurun): what bounds K2 `bpr_step_kernel` at batch 8192 / 65,536.
# One rocprofv3 pass per counter group (kernel trace only, as the pool requires); summarise with scripts/summarize_k2_counters.py.
set -u
TAG=${1:-r05}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/k2c_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in 8192 65536; do
  if [ $B = 8192 ]; then ST="--steps 128 --warmup 128"; else ST="--steps 32 --warmup 16"; fi
  for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
    N=$(echo $C | tr ' ' '_')
    timeout 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/B${B}_$N" -o b -- \
        python "$REPO/bench.py" --no-cpu-baseline --no-extras --batch-size $B $ST > /dev/null 2> "$OUT/B${B}_$N.err" < /dev/null
  done
done
find "$OUT" -type f \( -name "*.db" -o -name "*agent_info*" \) -delete
find "$OUT" -name "*.csv" -size +8M -exec gzip -9 {} \;
du -sh "$OUT"
