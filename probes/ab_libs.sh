#!/bin/bash
# same-box A/B of two builds of the library: scratch/lib_head.so vs scratch/lib_new.so, alternating
for rep in 1 2 3; do
  for v in head new; do
    cp scratch/lib_$v.so texocr_amd/libtexocr_hip.so
    echo -n "$v: "; python probes/pbench.py one 2>&1 | tail -1
  done
done
