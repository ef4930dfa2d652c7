#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p0 -- python3 scratch/enc_only.py > $O/p0.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- python3 scratch/enc_only.py > $O/p1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p2 -- python3 scratch/enc_only.py > $O/p2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_attn/p0/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us {r["Percentage"]}%')
for p in ("p1", "p2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_attn/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "enc_attn" in n:
                acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(p, k)
        for cn, v in sorted(c.items()):
            print(f"     {cn:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
