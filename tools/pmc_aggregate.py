"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files: pmc_aggregate.py DIR [name-filter]."""
import csv, sys, glob, collections, re
# usage: pmcagg.py dir [name-filter]
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
seen = set()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(.*', '', r['Kernel_Name'])
        k = re.sub(r'^void ', '', k)
        if flt and flt not in k: continue
        key = k + ' g' + r['Grid_Size'] + ' v' + r['VGPR_Count'] + ' lds' + r['LDS_Block_Size']
        acc[key][r['Counter_Name']] += float(r['Counter_Value'])
        if (r['Dispatch_Id']) not in seen:
            seen.add(r['Dispatch_Id']); cnt[key] += 1; dur[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k in sorted(acc, key=lambda k: -dur[k])[:12]:
    n = cnt[k]
    print(f'{k}  launches={n} avg_us={dur[k]/n/1e3:.1f}')
    for c, v in sorted(acc[k].items()): print(f'    {c:28s} {v/n:16.1f}')
