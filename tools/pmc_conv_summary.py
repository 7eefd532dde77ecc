import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        kn = r['Kernel_Name']
        if 'conv_' not in kn:
            continue
        key = (kn.split('(')[0][-60:], r['Grid_Size'], r['LDS_Block_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, c in agg.items():
    print(key)
    for k, v in sorted(c.items()):
        print('   %-34s %16.0f' % (k, sum(v) / len(v)))
