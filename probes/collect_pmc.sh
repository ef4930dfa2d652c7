#!/bin/bash
# HBM traffic of the decode kernels from the PMC counters, one counter per rocprofv3 pass (never combined with trace
# domains other than --kernel-trace).  Run on the GPU box from the repository root:  bash probes/collect_pmc.sh bf16 64 [max_len]
# then  python probes/pmc_summary.py gpurun_out/pmc_bf16_b64 bf16 64 [max_len] > profiles/r02_pmc_bf16_b64.json
# Launch-per-stage kernels: TXO_PERSIST=0 and a short decode (24 positions) are enough -- traffic per launch does not depend on
# the position for the cross-attention kernel.  Persistent decode launch: pass max_len 256 (its traffic is the whole loop's).
dt=${1:-bf16}
b=${2:-64}
ml=${3:-24}
tag=pmc_${dt}_b${b}
if [ "$ml" != "24" ]; then export TXO_PERSIST=1; tag=pmc_persist_${dt}_b${b}; else export TXO_PERSIST=0; export TXO_LANES=1; fi   # launches: ONE row range, a launch covers the whole batch
[ -n "$TXO_PMC_TAG" ] && tag=${tag}_$TXO_PMC_TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/${tag}/$c -- \
      python3 bench.py --steps 1 --warmup 0 --settle-seconds 0 --max-len $ml --dtype $dt --batch $b --no-cpu-baseline --no-roofline --no-extras \
      > gpurun_out/${tag}_$c.log 2>&1
done
