cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
FL="--steps 2 --warmup 1 --settle-seconds 0 --no-extras --no-cpu-baseline --no-roofline --batch 256 --model cfg4"
TXO_LANES=1 timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg4_one -- python3 bench.py $FL > $O/prof_cfg4_one.log 2>&1
f=$(find $O/prof_cfg4_one -name "*kernel_stats.csv" | head -1); cp "$f" $O/r04_cfg4_one_range_bf16_kernel_stats.csv
find $O/prof_cfg4_one -name "*kernel_trace.csv" -delete; find $O/prof_cfg4_one -name "*.db" -delete
tail -2 $O/prof_cfg4_one.log
