#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv: share, calls, average us."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    share = float(r["TotalDurationNs"]) / tot * 100
    print(f"{share:5.1f}%  calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  {r['Name'][:120]}")
