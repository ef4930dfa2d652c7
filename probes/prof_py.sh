#!/bin/bash
# per-kernel profile of a python probe: probes/prof_py.sh <tag> <script.py> [args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; tag=$1; shift
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 "$@" > $O/prof_$tag.log 2>&1
f=$(find $O/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
find $O/prof_$tag -name "*kernel_trace.csv" -delete; find $O/prof_$tag -name "*.db" -delete
python3 - "$O/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>7} avg {float(r['AverageNs'])/1e3:9.2f} us  {float(r['Percentage']):5.1f} %")
PY
