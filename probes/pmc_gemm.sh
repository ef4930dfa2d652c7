#!/bin/bash
# SQ / TCC counters of the ViT-Base encoder's kernels (BASELINE cfg 4, B=256): separate rocprofv3 --pmc passes (--kernel-trace only), per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
pass() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 probes/enc_prof.py > $O/$n.log 2>&1; }
pass p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass p2 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA
pass p3 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass p4 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3", "p4"):
    for f in glob.glob(f"gpurun_out/pmc_gemm/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm_pp" in n or "enc_attn" in n or "ln_rows" in n:
                key = n.split("(")[0].replace("void txo::", "")[:70]
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for cn, v in sorted(c.items()):
        print(f"     {cn:30s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
