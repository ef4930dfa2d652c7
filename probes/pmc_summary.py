"""Summarise the rocprofv3 --pmc passes of probes/collect_pmc.sh: average FETCH_SIZE / WRITE_SIZE (KB) per launch of every kernel and
the corrected HBM bytes of the cross-attention launch (gfx950 reports half the bytes of 16-B/lane streaming reads in FETCH_SIZE:
MI355X_MICROARCH.md, HBM section -> read bytes = 2 * FETCH_SIZE * 1024).  python probes/pmc_summary.py <dir> <dtype> [batch]"""
import re
import csv, glob, json, sys
from collections import defaultdict

root, dtype = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]][counter].append(float(r["Counter_Value"]))
B, heads, N, esz = (int(sys.argv[3]) if len(sys.argv) > 3 else 64), 8, 589, (2 if dtype == "bf16" else 4)
max_len = int(sys.argv[4]) if len(sys.argv) > 4 else 24
# rows ONE profiled launch covers: the profiled run decodes the batch as `ranges` row ranges (collect_pmc.sh runs with TXO_LANES=1, so
# a launch covers the whole batch; r03's passes ran the two-range default and their 128-row launches were labelled with 256-row bytes)
ranges = int(sys.argv[5]) if len(sys.argv) > 5 else 1
rows_per_launch = B // ranges
D = 256
kernels, cross, persist = {}, None, None
for name, c in sorted(acc.items()):
    if "txo::" not in name:
        continue
    e = {"FETCH_SIZE_KB_avg": round(sum(c["FETCH_SIZE"]) / max(1, len(c["FETCH_SIZE"])), 1),
         "WRITE_SIZE_KB_avg": round(sum(c["WRITE_SIZE"]) / max(1, len(c["WRITE_SIZE"])), 1), "launches": len(c["FETCH_SIZE"])}
    if "dec_attn_kernel" in name and re.search(r"dec_attn_kernel<[^,]+, 0, 1, \d+,", name):   # cross attention (MODE 0, APRO_LN2, any NL)
        e["algorithmic_bytes_per_launch"] = rows_per_launch * heads * 2 * N * 64 * esz
        e["rows_per_launch"] = rows_per_launch
        e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KB_avg"] * 1024 + e["WRITE_SIZE_KB_avg"] * 1024)
        cross = (name, e)
    if "lat_core_kernel" in name:                                          # cross attention in latent form: the raw encoder rows once for all heads
        e["algorithmic_bytes_per_launch"] = rows_per_launch * N * D * esz
        e["algorithmic_bytes_kv_form"] = rows_per_launch * heads * 2 * N * 64 * esz
        e["rows_per_launch"] = rows_per_launch
        e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KB_avg"] * 1024 + e["WRITE_SIZE_KB_avg"] * 1024)
        cross = (name, e)
    if "decode_persist_kernel" in name:                                    # the whole decode loop as one launch
        rows = sum(2 * N + 2 * (t + 1) for t in range(max_len))
        e["algorithmic_bytes_per_launch"] = B * heads * 64 * esz * 4 * rows     # 4 decoder layers
        e["hbm_bytes_per_launch_corrected"] = int(2 * e["FETCH_SIZE_KB_avg"] * 1024 + e["WRITE_SIZE_KB_avg"] * 1024)
        persist = (name, e)
    kernels[name] = e
if persist is not None:
    print(json.dumps({"note": "rocprofv3 --pmc <counter> --kernel-trace, separate passes (probes/collect_pmc.sh <dtype> <batch> <max_len>): the "
                              f"persistent decode launch, B={B}, {dtype}, 3x224x672, {max_len} positions.  hbm_read_bytes = 2 * FETCH_SIZE * 1024 "
                              "(gfx950 correction for 16-byte-per-lane streams; the launch also reads weights and rows through L2 with narrower "
                              "accesses, for which the factor is uncalibrated).",
                      "traffic": {"config": {"batch": B, "dtype": dtype, "tokens": N, "max_len": max_len}, "kernel": persist[0],
                                  "traffic_bytes": persist[1]["hbm_bytes_per_launch_corrected"],
                                  "algorithmic_bytes": persist[1]["algorithmic_bytes_per_launch"]},
                      "kernels": kernels}, indent=1, sort_keys=True))
    sys.exit(0)
out = {"note": "rocprofv3 --pmc <counter> --kernel-trace, separate passes (probes/collect_pmc.sh), bench.py --steps 1 --max-len 24 "
               f"(B={B}, {dtype}, 3x224x672). FETCH_SIZE/WRITE_SIZE are in KB; hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950 correction).",
       "cross_attention_traffic": {"config": {"batch": B, "dtype": dtype, "tokens": N}, "kernel": cross[0], "rows_per_launch": rows_per_launch,
                                   "traffic_bytes": cross[1]["hbm_bytes_per_launch_corrected"],
                                   "algorithmic_bytes": cross[1]["algorithmic_bytes_per_launch"]},
       "kernels": kernels}
print(json.dumps(out, indent=1, sort_keys=True))
