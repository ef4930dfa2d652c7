#!/bin/bash
# HBM traffic of the decode kernels from the PMC counters, one counter per rocprofv3 pass (never combined with trace
# domains other than --kernel-trace).  Run on the GPU box from the repository root:  bash probes/collect_pmc.sh bf16 64
# then  python probes/pmc_summary.py gpurun_out/pmc_bf16_b64 bf16 64 > profiles/r02_pmc_bf16_b64.json
dt=${1:-bf16}
b=${2:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${dt}_b${b}/$c -- \
      python3 bench.py --steps 1 --warmup 0 --settle-seconds 0 --max-len 24 --dtype $dt --batch $b --no-cpu-baseline --no-roofline --no-extras \
      > gpurun_out/pmc_${dt}_b${b}_$c.log 2>&1
done
