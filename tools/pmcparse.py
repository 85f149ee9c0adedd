import csv, collections, sys
tag = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'k_probe'
rows = list(csv.DictReader(open(f"gpurun_out/{tag}/{tag}_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
name = {}
for r in rows:
    if pat in r['Kernel_Name'] and 'mem' not in r['Kernel_Name']:
        d = r['Dispatch_Id']
        name[d] = r['Kernel_Name'][:40]
        agg[d][r['Counter_Name']] += float(r['Counter_Value'])
        agg[d]['dur_us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: int(kv[0]))[-2:]:
    print(name[k], {a: round(b) for a, b in v.items()})
