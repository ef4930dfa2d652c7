#!/bin/bash
# same-box A/B of library builds on the latent / K-V persistent decode:  probes/ab_latent.sh B lib_a.so lib_b.so ...
B=$1; shift
cp texocr_amd/libtexocr_hip.so /tmp/lib_default.so
for rep in 1 2; do
  for v in "$@"; do
    cp "$v" texocr_amd/libtexocr_hip.so
    echo "== $v"; python probes/latent_pbench.py $B 2>&1 | grep -v amdgpu.ids | head -2
  done
done
cp /tmp/lib_default.so texocr_amd/libtexocr_hip.so
