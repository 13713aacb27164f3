#!/usr/bin/env python3
"""Average the counters of one kernel from a rocprofv3 --pmc counter_collection.csv: pmc_kernel.py FILE.csv KERNEL_SUBSTRING"""
import csv, statistics, sys
vals = {}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(vals.items()):
    print(f"{k:<32} mean {statistics.mean(v):16.1f}  (n={len(v)})")
