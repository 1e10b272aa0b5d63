import csv, glob, collections, sys
tag = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'pf_kernel'
for d in sorted(glob.glob('gpurun_out/pmc_%s_*' % tag)):
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if pat in k:
                print(k)
                for c, vals in sorted(v.items()):
                    print('   %-28s %.4g' % (c, sum(vals) / len(vals)))
