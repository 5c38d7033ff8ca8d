#!/bin/bash
# Run ON THE GPU BOX (through gpurun): matrix-pipe counters of the MFMA kernels (north_star: "MFMA-busy counters against gfx950 peak").
#   one rocprofv3 pass per counter (kernel trace only, as the pool requires) over
#     topk     K4 bound-and-refine (fp16 MFMA), ML-10M shape      topknf   the same, Netflix shape
#     topk32   K4 fp32-MFMA kernel (TKR_TOPK_MATH=fp32), ML-10M shape
#     vbprd    VBPR dense view (fp32 MFMA: vbpr_project / vbpr_dense), d = 20,000, 16 batches
# Output: gpurun_out/mfma_<tag>/<leg>_<COUNTER>/...; summarise with `python scripts/summarize_mfma.py gpurun_out/mfma_<tag> <tag>`.
set -u
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/mfma_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES; do
    REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/topk_$C" -o b -- \
        python "$REPO/scripts/probe_topk.py" > "$OUT/topk_$C.out" 2> "$OUT/topk_$C.err" < /dev/null
    REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/topknf_$C" -o b -- \
        python "$REPO/scripts/probe_topk.py" 480189 17770 128 30 > "$OUT/topknf_$C.out" 2> "$OUT/topknf_$C.err" < /dev/null
    REPS=2 TKR_TOPK_MATH=fp32 timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/topk32_$C" -o b -- \
        python "$REPO/scripts/probe_topk.py" > "$OUT/topk32_$C.out" 2> "$OUT/topk32_$C.err" < /dev/null
    NB=16 VIEW=dense timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/vbprd_$C" -o b -- \
        python "$REPO/scripts/probe_vbpr.py" > "$OUT/vbprd_$C.out" 2> "$OUT/vbprd_$C.err" < /dev/null
done
find "$OUT" -type f \( -name "*.db" -o -name "*agent_info*" \) -delete
find "$OUT" -name "*.csv" -size +8M -exec gzip -9 {} \;
du -sh "$OUT"
