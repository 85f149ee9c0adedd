import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(f"gpurun_out/{tag}/{tag}_kernel_trace.csv")))
rows = [r for r in rows if "k_gen" not in r["Kernel_Name"] and "k_walk" not in r["Kernel_Name"]]
rows = rows[len(rows) // 2:]                      # steady state
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = 0; depth = 0; last = ev[0][0]; maxd = 0; wsum = 0
for t, d in ev:
    if depth > 0: busy += t - last
    wsum += depth * (t - last)
    last = t; depth += d; maxd = max(maxd, depth)
span = ev[-1][0] - ev[0][0]
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print(f"kernels {len(rows)} span {span/1e6:.2f} ms, GPU busy {busy/span:.2%}, sum of durations / span {tot/span:.2f}, max concurrent {maxd}")
qs = {}
for r in rows:
    qs.setdefault(r.get("Queue_Id", "?"), 0); qs[r.get("Queue_Id", "?")] += 1
print("queues:", qs)
