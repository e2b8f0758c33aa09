"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, sum / mean of a counter."""
import csv
import glob
import os
import sys
from collections import defaultdict


def summarise(d):
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name', '?')[:90]
                acc[name][row.get('Counter_Name', '?')].append(float(row.get('Counter_Value', 0)))
        print('#', f)
        for name, cs in sorted(acc.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values()))[:12]:
            for c, vals in cs.items():
                print('%-90s %-12s n=%-5d mean=%.6g sum=%.6g' % (name, c, len(vals), sum(vals) / len(vals), sum(vals)))


for d in sys.argv[1:]:
    summarise(d)
