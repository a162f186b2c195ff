"""print per-kernel ms/step from a rocprofv3 --stats kernel_stats.csv (usage: kstats.py <csv> <forwards in the run>)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / n / 1e6:.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 34]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:8.2f} ms/step  calls/step {int(r['Calls']) / n:7.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:140]}")
