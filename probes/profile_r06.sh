#!/bin/bash
# Round-6 evidence run (GPU box, repository root).  Outputs under gpurun_out/; the summaries are copied to profiles/ by hand.
#  * the default bench line, and the same under torch.distributed.run with one rank (RCCL all-gather inside the timed region)
#  * rocprofv3 --kernel-trace --stats: batch 64 (persistent launch), batch 256 on its default path (two row ranges, latent cross attention),
#    batch 256 as ONE row range (a launch covers the whole batch: the lat_core row the bench's b256.roofline is computed from), the same in
#    the K/V form (TXO_LATENT=0: the ~48 us whole-batch dec_attn launch of rounds 1-3), config 4
#  * PMC traffic (one counter per pass): the persistent launch at batch 64; lat_core and the K/V cross-attention launch at batch 256, one row range
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
python3 bench.py > $O/r06_bench_default.json 2> $O/r06_bench_default.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 \
    --no-extras --no-cpu-baseline > $O/r06_bench_torchrun1.json 2> $O/r06_bench_torchrun1.err
FL="--steps 4 --warmup 1 --settle-seconds 0 --no-extras --no-cpu-baseline --no-roofline"
prof() { tag=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_$tag -- python3 bench.py $FL "$@" > $O/prof_r06_$tag.log 2>&1;
         f=$(find $O/prof_r06_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_${tag}_bf16_kernel_stats.csv; }
prof b64
prof b256 --batch 256
TXO_LANES=1 prof b256_one_range --batch 256
TXO_LANES=1 TXO_LATENT=0 TXO_PERSIST=0 prof b256_one_range_kvform --batch 256
prof cfg4 --batch 256 --model cfg4 --steps 2
TXO_LANES=1 prof cfg4_one_range --batch 256 --model cfg4 --steps 2
# beam search at the BASELINE configs[4] shape (128 images x 5 beams, 224x672): two row ranges (default) and one
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r06_cfg5 -- python3 probes/cfg5_prof.py > $O/prof_r06_cfg5.log 2>&1
f=$(find $O/prof_r06_cfg5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r06_cfg5_beam_final_kernel_stats.csv
bash probes/collect_pmc.sh bf16 64 256   # (profiles/r06_pmc_*.json; bench.py reads the newest round it finds)
TXO_PMC_TAG=latent bash probes/collect_pmc.sh bf16 256
TXO_PMC_TAG=kvform TXO_LATENT=0 bash probes/collect_pmc.sh bf16 256
python3 probes/pmc_summary.py $O/pmc_persist_bf16_b64 bf16 64 256 > $O/r06_pmc_persist_bf16_b64.json
python3 probes/pmc_summary.py $O/pmc_bf16_b256_latent bf16 256 24 1 > $O/r06_pmc_bf16_b256.json
python3 probes/pmc_summary.py $O/pmc_bf16_b256_kvform bf16 256 24 1 > $O/r06_pmc_bf16_b256_kvform.json
python3 probes/sample_bench.py > $O/r06_sampled_decode.txt 2>&1
# gpurun copies back at most 64 MiB: the raw traces and counter dumps are not needed once the summaries exist
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
ls -la $O | grep r06 | tail -24
# round 6 extras: the ViT-Base encoder alone (per-kernel), the opt-in self attention on the z history
bash probes/prof_py.sh r06_cfg4_encoder probes/enc_prof.py > /dev/null 2>&1
# the default-factory (hybrid) model's encoder, batch 64, and the row-stop / hybrid bench extras on their own
bash probes/prof_py.sh r06_hybrid probes/hyb_prof.py bf16 64 > /dev/null 2>&1
python3 probes/row_stop_bench.py 256 90 > $O/r06_row_stop_b256.json 2>/dev/null
ls -la $O | grep r06 | tail -40
