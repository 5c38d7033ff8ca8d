#!/bin/bash
# run ON THE GPU BOX: per-batch time of the column-plan VBPR step with pieces switched off (TKR_VBPR_TUNE bits, csrc/vbpr_cols.hip)
cd /tmp && export TMPDIR=/tmp
for T in 0 1 2 3 4 8 12 16 28 32 64 128 96 160 192; do
  echo -n "tune=$T  "; TKR_VBPR_TUNE=$T NB=${NB:-512} python /root/repo/scripts/probe_vbpr.py 2>&1 | tail -1
done
