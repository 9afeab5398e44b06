"""Short view of a rocprofv3 *_kernel_stats.csv: python benchmarks/kstats.py FILE [rows]"""
import csv
import sys

rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= rows:
        break
    print(f"{r['Name'][:44]:44s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  "
          f"max {float(r['MaxNs']) / 1e3:9.1f}  {float(r['Percentage']):5.1f} %")
