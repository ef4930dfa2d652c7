#!/bin/bash
# same-box A/B of library builds on the encoder alone: probes/ab_enc.sh lib_a.so lib_b.so ...   (TXO_ENC_NT passes through)
cp texocr_amd/libtexocr_hip.so /tmp/lib_default.so
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" texocr_amd/libtexocr_hip.so
    for nt in 0 1; do echo -n "$v TXO_ENC_NT=$nt: "; TXO_ENC_NT=$nt python probes/enc_one.py 2>&1 | tail -1; done
  done
done
cp /tmp/lib_default.so texocr_amd/libtexocr_hip.so
