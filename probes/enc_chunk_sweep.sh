#!/bin/bash
# r06: ViT-Base encoder (BASELINE cfg 4, B=256) in image chunks -- per-kernel time per ENCODE for every chunk size (rocprofv3 --kernel-trace --stats).
#   probes/enc_chunk_sweep.sh "0 32 64 128"      (0 = whole batch)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for c in ${1:-0 32 64 128}; do
  export TXO_ENC_CHUNK=$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_encchunk$c -- python3 probes/enc_prof.py > $O/prof_encchunk$c.log 2>&1
  f=$(find $O/prof_encchunk$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_enc_chunk${c}_kernel_stats.csv
  rm -rf $O/prof_encchunk$c
done
python3 - $O ${1:-0 32 64 128} <<'PY'
import csv, sys, re
O, chunks = sys.argv[1], sys.argv[2:]
ENCODES = 4                                   # probes/enc_prof.py runs four encodes
def short(n):
    n = re.sub(r"^void txo::", "", n)
    m = re.match(r"(\w+)<(.*)>\(", n)
    return (m.group(1) + "<" + m.group(2)[:60] + ">") if m else n[:80]
tab, names = {}, []
for c in chunks:
    for r in csv.DictReader(open(f"{O}/r06_enc_chunk{c}_kernel_stats.csv")):
        k = short(r["Name"])
        if k not in names: names.append(k)
        tab[(k, c)] = (int(r["Calls"]) // ENCODES, float(r["TotalDurationNs"]) / ENCODES / 1e3)
print("per-kernel total us per encode (launches per encode), chunk = images per pass through the 12 layers; 0 = whole batch of 256")
print(f"{'kernel':90s}" + "".join(f"{'chunk ' + c:>20s}" for c in chunks))
tot = {c: 0.0 for c in chunks}
for k in names:
    line = f"{k:90s}"
    for c in chunks:
        n, us = tab.get((k, c), (0, 0.0)); tot[c] += us
        line += f"{us:12.1f} ({n:5d})"
    print(line)
print(f"{'sum of kernel time':90s}" + "".join(f"{tot[c]:12.1f}        " for c in chunks))
PY
