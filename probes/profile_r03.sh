#!/bin/bash
# Round-3 evidence run (GPU box, repository root): the default bench line, the same under torch.distributed.run with one rank
# (RCCL all-gather inside the timed region), a sampled-decode line (the reference's default decode on the persistent launch),
# rocprofv3 --kernel-trace --stats summaries for batch 64 / batch 256 / config 4 / sampled decode, and the PMC traffic passes
# (one counter per pass) for the persistent launch and the launch-per-stage cross-attention kernel.
# Outputs under gpurun_out/; the summaries are copied to profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/r03_bench_default.json 2> $O/r03_bench_default.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 \
    --no-extras --no-cpu-baseline > $O/r03_bench_torchrun1.json 2> $O/r03_bench_torchrun1.err
FL="--steps 4 --warmup 1 --settle-seconds 0 --no-extras --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_b64 -- python3 bench.py $FL > $O/prof_r03_b64.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_b256 -- python3 bench.py $FL --batch 256 > $O/prof_r03_b256.log 2>&1
export TXO_PERSIST=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_b64_launches -- python3 bench.py $FL > $O/prof_r03_b64_launches.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_b256_launches -- python3 bench.py $FL --batch 256 > $O/prof_r03_b256_launches.log 2>&1
unset TXO_PERSIST
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_cfg4 -- python3 bench.py $FL --batch 256 --model cfg4 --steps 2 > $O/prof_r03_cfg4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_prefill -- python3 probes/prefill_bench.py > $O/prof_r03_prefill.log 2>&1
bash probes/collect_pmc.sh bf16 64 256
bash probes/collect_pmc.sh bf16 64
bash probes/collect_pmc.sh bf16 256
python3 probes/pmc_summary.py $O/pmc_persist_bf16_b64 bf16 64 256 > $O/r03_pmc_persist_bf16_b64.json
python3 probes/pmc_summary.py $O/pmc_bf16_b64 bf16 64 > $O/r03_pmc_bf16_b64.json
python3 probes/pmc_summary.py $O/pmc_bf16_b256 bf16 256 > $O/r03_pmc_bf16_b256.json
for d in b64 b256 cfg4 b64_launches b256_launches prefill; do f=$(find $O/prof_r03_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_${d}_bf16_kernel_stats.csv; done
python3 probes/sample_bench.py > $O/r03_sampled_decode.txt 2>&1
# gpurun copies back at most 64 MiB: the raw traces and counter dumps are not needed once the summaries exist
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O | tail -24
