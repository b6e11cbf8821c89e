import csv, sys, collections, re
f = sys.argv[1]; steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    n = r["Name"]
    n = re.sub(r"\(.*", "", n)[:70]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{t/1e6/steps:8.3f} ms/step  {c/steps:7.1f} calls  {t/c/1e3:8.1f} us  {n}")
print(f"total {tot/1e6/steps:.2f} ms/step")
