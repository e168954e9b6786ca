import csv, collections, sys
for f in sys.argv[1:]:
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        k=r['Kernel_Name'][:48]
        if 'gemm' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        print(k)
        for c,x in sorted(v.items()): print('   %-32s %.4e'%(c,x))
