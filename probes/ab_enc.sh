#!/bin/bash
# same-box A/B of library builds on the encoder alone: probes/ab_enc.sh lib_a.so lib_b.so ...   (TXO_ENC_NT passes through; the in-tree
# product library is not touched: TXO_LIB_PATH, texocr_amd/_lib.py)
for rep in 1 2; do
  for v in "$@"; do
    for nt in 0 1; do echo -n "$v TXO_ENC_NT=$nt: "; TXO_LIB_PATH="$(readlink -f "$v")" TXO_ENC_NT=$nt python probes/enc_one.py 2>&1 | tail -1; done
  done
done
