"""Rows of a rocprofv3 counter_collection.csv whose kernel name contains argv[2] (keeps the committed PMC files small)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0].keys()))
w.writeheader()
for r in rows:
    if sys.argv[2] in r['Kernel_Name']:
        w.writerow(r)
