"""Sum rocprofv3 --pmc counter_collection.csv per kernel and counter."""
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(path)):
        k = (row['Kernel_Name'][:40], row['Counter_Name'])
        tot[k] += float(row['Counter_Value']); cnt[k] += 1
for k in sorted(tot):
    print(f"{k[0]:40s} {k[1]:28s} sum={tot[k]:.4g} dispatches={cnt[k]} per_dispatch={tot[k]/cnt[k]:.4g}")
