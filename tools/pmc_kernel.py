"""Average PMC counters of the dispatches whose kernel name contains <needle>: python tools/pmc_kernel.py <dir> <needle>"""
import csv, glob, json, sys
from collections import defaultdict
d, needle = sys.argv[1:3]
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if needle in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print(json.dumps({k: acc[k] / n[k] for k in acc}, indent=0))
