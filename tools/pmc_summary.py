#!/usr/bin/env python
"""Collapse rocprofv3 --pmc counter_collection CSVs under <dir>/pmc*/ into kernel,counter,dispatches,mean_per_dispatch."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/pmc*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:72], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel,counter,dispatches,mean_per_dispatch")
for (k, c), v in sorted(acc.items()):
    print('"%s",%s,%d,%.3f' % (k, c, len(v), sum(v) / len(v)))
