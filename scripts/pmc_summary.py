import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows: agg[r['Kernel_Name'][:46]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'rtuf' not in k: continue
    print(k)
    for c,vals in sorted(v.items()): print('     %-26s avg %.5g  (n=%d)'%(c,sum(vals)/len(vals),len(vals)))
