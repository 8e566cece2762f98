"""Mean of every counter per (kernel, grid size) from a rocprofv3 counter_collection.csv; argv: csv, kernel-name substring."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, g), cs in sorted(acc.items()):
    print(k, 'grid', g)
    for c, v in sorted(cs.items()):
        print(f'    {c:32s} n={len(v):3d} mean={sum(v) / len(v):.4g}')
