import csv, collections, sys
tag=sys.argv[1]
rows=list(csv.DictReader(open(f"gpurun_out/{tag}/{tag}_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if 'k_probe' in r['Kernel_Name'] and 'mem' not in r['Kernel_Name']:
        agg[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
        agg[r['Dispatch_Id']]['dur_us']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        agg[r['Dispatch_Id']]['vgpr']=float(r['VGPR_Count']); agg[r['Dispatch_Id']]['lds']=float(r['LDS_Block_Size'])
for k,v in sorted(agg.items(), key=lambda kv:int(kv[0]))[-1:]: print({a:round(b) for a,b in v.items()})
