#!/bin/bash
# Builds libtkr_hip variants with parts of the K4 tile loop compiled out (TKR_ABL bits: 1 filter, 2 staging, 4 barrier) into
# _ab_libs/ -- timing experiments only, results are wrong by construction.  Run on the build host; then on the GPU box:
#   for a in 0 1 3 7; do TKR_HIP_LIB=$PWD/_ab_libs/libtkr_abl$a.so python scripts/probe_topk_modes.py netflix refine; done
set -e
cd "$(dirname "$0")/../top-k-rec_amd/csrc"
mkdir -p ../../_ab_libs
for a in ${1:-0 1 3 7}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DTKR_ABL=$a -c topk.hip -o /tmp/topk_abl$a.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/*.o | grep -v topk.o) /tmp/topk_abl$a.o -o ../../_ab_libs/libtkr_abl$a.so
done
