#!/bin/bash
# register / scratch use of the library's kernels without a GPU: device-only assembly of engine.hip, then the code-object
# metadata (.vgpr_count, .private_segment_fixed_size = scratch bytes, .sgpr_count, LDS) of every kernel matching $1 (regex)
set -e
OUT=${TXO_KSTATS_ASM:-/tmp/txo_engine_gfx950.s}
HERE=$(cd "$(dirname "$0")/.." && pwd)
if [ ! -f "$OUT" ] || [ -n "$(find "$HERE/texocr_amd/csrc" -newer "$OUT" -name '*.h*' | head -1)" ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=on -fno-honor-nans \
      ${TXO_EXTRA_FLAGS} "$HERE/texocr_amd/csrc/engine.hip" -o "$OUT" 2>/dev/null
fi
python3 - "$OUT" "${1:-.}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if not pat.search(name): continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print(f"{name[:110]:110s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6}")
PY
