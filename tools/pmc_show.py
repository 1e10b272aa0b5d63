import csv, glob, collections, sys
tag = sys.argv[1]
for d in sorted(glob.glob('gpurun_out/pmc_%s_*' % tag)):
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:44]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in agg.items():
            if 'pf_kernel' in k:
                for c, vals in sorted(v.items()):
                    print('   %-28s %.4g' % (c, sum(vals) / len(vals)))
