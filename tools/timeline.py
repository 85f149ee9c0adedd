import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}/{tag}_kernel_trace.csv")))
idx = [i for i, r in enumerate(rows) if "k_make_keys" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(r["Kernel_Name"][:70].ljust(70), round((int(r["Start_Timestamp"]) - t0) / 1e3, 1), round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1))
