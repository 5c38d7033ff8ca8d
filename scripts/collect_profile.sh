#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for bench.py.
#   trace            kernel trace + stats of the full bench (every leg)
#   headline_C       separate PMC passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the pool requires) of the HEADLINE leg alone
#                    (--no-extras --steps 2048 --warmup 256: 2,304 batches of 256 through the persistent kernel)
#   b8192_C          the same for the per-launch kernel at batch_size 8192 (--batch-size 8192 --steps 256 --warmup 128)
#   calib_C          the known-byte calibration copy
#   driver           HIP API + kernel trace of the round-end driver's exact command (--steps 20 --warmup 5): what the timed region calls
# Output: gpurun_out/prof_<tag>/...; summarise afterwards with `python scripts/summarize_profile.py gpurun_out/prof_<tag> <tag>`.
set -u
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" < /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/headline_$C" -o b -- \
        python "$REPO/bench.py" --no-cpu-baseline --no-extras --steps 2048 --warmup 256 > /dev/null 2> "$OUT/headline_$C.err" < /dev/null
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/b8192_$C" -o b -- \
        python "$REPO/bench.py" --no-cpu-baseline --no-extras --batch-size 8192 --steps 256 --warmup 128 > /dev/null 2> "$OUT/b8192_$C.err" < /dev/null
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/calib_$C" -o c -- \
        python "$REPO/scripts/pmc_calibrate.py" > /dev/null 2> "$OUT/calib_$C.err" < /dev/null
    # round 3: K4's bound-and-refine kernel (ML-10M and Netflix shape), the per-launch kernel at batch 65,536, the VBPR step
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/topk_$C" -o b -- \
        python "$REPO/scripts/probe_topk.py" > /dev/null 2> "$OUT/topk_$C.err" < /dev/null
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/topknf_$C" -o b -- \
        python "$REPO/scripts/probe_topk.py" 480189 17770 128 30 > /dev/null 2> "$OUT/topknf_$C.err" < /dev/null
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/b65536_$C" -o b -- \
        python "$REPO/bench.py" --no-cpu-baseline --no-extras --batch-size 65536 --steps 32 --warmup 16 > /dev/null 2> "$OUT/b65536_$C.err" < /dev/null
    NB=256 timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/vbpr_$C" -o b -- \
        python "$REPO/scripts/probe_vbpr.py" > /dev/null 2> "$OUT/vbpr_$C.err" < /dev/null
done
timeout 600 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d "$OUT/driver" -o d -- \
    python "$REPO/bench.py" --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > "$OUT/driver_cmd.json" 2> "$OUT/driver.err" < /dev/null
# the merged-back directory is capped at 64 MiB: keep the stats, the per-kernel trace and the counter tables only
find "$OUT" -name "*.csv" -size +20M -exec gzip -9 {} \;
find "$OUT" -type f \( -name "*.db" -o -name "*agent_info*" \) -delete
du -sh "$OUT"
find "$OUT" -name "*.csv*" | head -40
