#!/bin/bash
# Instruction-cache behaviour of the persistent decode launch (97.9 KB of code; the instruction cache is 64 KB per two CUs):
# SQC_ICACHE_REQ / _HITS / _MISSES and the instruction bytes fetched from L2 (SQC_TC_INST_REQ), one rocprofv3 pass.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export TXO_PERSIST=1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d gpurun_out/pmc_icache -- \
    python3 bench.py --steps 1 --warmup 0 --settle-seconds 0 --max-len 256 --dtype bf16 --batch 64 --no-cpu-baseline --no-roofline --no-extras > gpurun_out/pmc_icache.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmc_icache/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "persist" in k or "dec_" in k: print(k, dict(v))
PY
