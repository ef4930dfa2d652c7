#!/bin/bash
# same-box A/B of several BUILDS of the library, alternating three times:  probes/ab_libs.sh scratch/lib_a.so scratch/lib_b.so ...
# (build them with TXO_HIPCC_FLAGS=... TXO_LIB_OUT=scratch/lib_x.so python -m texocr_amd.build).  The in-tree product library is not
# touched: each run loads its build through TXO_LIB_PATH (texocr_amd/_lib.py).
for rep in 1 2 3; do
  for v in "$@"; do
    echo -n "$v: "; TXO_LIB_PATH="$(readlink -f "$v")" python probes/pbench.py one 2>&1 | tail -1
  done
done
