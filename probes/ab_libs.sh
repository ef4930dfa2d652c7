#!/bin/bash
# same-box A/B of several BUILDS of the library, alternating three times:  probes/ab_libs.sh scratch/lib_a.so scratch/lib_b.so ...
# (build them with TXO_HIPCC_FLAGS=... TXO_LIB_OUT=scratch/lib_x.so python -m texocr_amd.build); the default library is restored
cp texocr_amd/libtexocr_hip.so /tmp/lib_default.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp "$v" texocr_amd/libtexocr_hip.so
    echo -n "$v: "; python probes/pbench.py one 2>&1 | tail -1
  done
done
cp /tmp/lib_default.so texocr_amd/libtexocr_hip.so
