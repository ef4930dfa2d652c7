#!/bin/bash
# same-box A/B of library builds on the latent / K-V persistent decode:  probes/ab_latent.sh B lib_a.so lib_b.so ...
# (the in-tree product library is not touched: TXO_LIB_PATH, texocr_amd/_lib.py)
B=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v"; TXO_LIB_PATH="$(readlink -f "$v")" python probes/latent_pbench.py $B 2>&1 | grep -v amdgpu.ids | head -2
  done
done
