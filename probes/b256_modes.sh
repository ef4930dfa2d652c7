#!/bin/bash
# ten fresh processes per mode, interleaved
for i in 1 2 3 4 5 6; do
  for mode in "" "TXO_CU_SPLIT=16" "TXO_CU_SPLIT=20" "TXO_LANES=1"; do
    env $mode python probes/b256_modes.py 256 2>&1 | grep "^B="
  done
done
