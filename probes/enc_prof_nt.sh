#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
for nt in 0 1; do
  TXO_ENC_NT=$nt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_enc_nt$nt -- python3 probes/enc_prof.py > $O/prof_enc_nt$nt.log 2>&1
  f=$(find $O/prof_enc_nt$nt -name "*kernel_stats.csv" | head -1)
  echo "== TXO_ENC_NT=$nt"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(f"{r['Name'][:95]:95s} calls {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:9.2f} us  {float(r['Percentage']):5.1f} %")
PY
  find $O/prof_enc_nt$nt -name "*kernel_trace.csv" -delete; find $O/prof_enc_nt$nt -name "*.db" -delete
done
