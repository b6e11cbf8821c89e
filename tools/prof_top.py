import csv, glob, sys
f = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".csv") else glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r['Name'][:84]:84s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
