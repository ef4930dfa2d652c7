#!/bin/bash
# statistics of the batch-256 generate time per stream-pair policy: fresh processes, k extra streams each
for rep in 1 2 3 4 5 6; do
  for k in 0 3 5; do
    for mode in "TXO_TUNE_LANES=0" "TXO_TUNE_TRIAL=0" "TXO_TUNE_TRIAL=1"; do
      r=$(env $mode python probes/b256_queues.py $k 2>&1 | grep "extra streams" | sed 's/extra streams //')
      echo "[$mode] k=$r"
    done
  done
done
