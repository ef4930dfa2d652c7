#!/bin/bash
for i in 1 2 3; do
  for mode in "" "TXO_GRAPH=1" "TXO_GRAPH=1 TXO_LANES=1" "TXO_LANES=1"; do
    env $mode python probes/b256_modes.py 256 2>&1 | grep "^B=" | sed "s/^/[$mode] /"
  done
done
