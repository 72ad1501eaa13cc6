"""Kernel name, calls and average duration from a rocprofv3 *kernel_stats.csv under a directory.

    python tools/kstats.py <directory> [substring ...]
"""
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
want = sys.argv[2:]
for f in files[:1]:
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(fdtdhip::.*$", "", row["Name"].replace("void fdtdhip::", ""))
        if want and not any(w in name for w in want):
            continue
        print(f"{name[:70]:70s} calls {row['Calls']:>6s}  avg {float(row['AverageNs']) / 1e3:9.1f} us  {row['Percentage']:>6s} %")
