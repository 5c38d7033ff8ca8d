#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + stats of bench.py, then separate PMC passes (FETCH_SIZE,
# WRITE_SIZE; kernel-trace only, as the pool requires) of bench.py and of the known-byte calibration copy.
# Output: gpurun_out/prof_<tag>/{trace,bench_FETCH_SIZE,bench_WRITE_SIZE,calib_FETCH_SIZE,calib_WRITE_SIZE}; summarise
# afterwards with `python scripts/summarize_profile.py gpurun_out/prof_<tag> <tag>`.
set -u
TAG=${1:-r01}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- \
    python "$REPO/bench.py" --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" < /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/bench_$C" -o b -- \
        python "$REPO/bench.py" --no-cpu-baseline --steps 2048 --warmup 256 > /dev/null 2> "$OUT/bench_$C.err" < /dev/null
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/calib_$C" -o c -- \
        python "$REPO/scripts/pmc_calibrate.py" > /dev/null 2> "$OUT/calib_$C.err" < /dev/null
done
# the merged-back directory is capped at 64 MiB: keep the stats, the per-kernel trace and the counter tables only
find "$OUT" -name "*.csv" -size +30M -exec gzip -9 {} \;
find "$OUT" -type f \( -name "*.db" -o -name "*agent_info*" \) -delete
du -sh "$OUT"
find "$OUT" -name "*.csv*" | head -20
