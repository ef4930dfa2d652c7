import csv, glob, sys
f = (glob.glob(sys.argv[1] + "/*kernel_trace.csv") + glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last argmax and print the 30 kernels before it (one steady-state step late in the run)
idx = [i for i, r in enumerate(rows) if "argmax" in r["Kernel_Name"]]
end = idx[-10]; start = idx[-11] + 1
prev_end = int(rows[start - 1]["End_Timestamp"])
tot = 0
for r in rows[start:end + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void txo::", "")[:58]
    print(f"gap {(s - prev_end)/1e3:6.2f} us  dur {(e - s)/1e3:6.2f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7} {name}")
    prev_end = e
print("step span us:", (int(rows[end]["End_Timestamp"]) - int(rows[start - 1]["End_Timestamp"])) / 1e3)
