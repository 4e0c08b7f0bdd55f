import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))
t0 = ev[0][0]
heads = [(s, e) for s, e, n in ev if "link_head" in n]
print("link_head starts (ms) and deltas:")
print(" ".join("%.1f" % ((b[0] - a[0]) / 1e6) for a, b in zip(heads, heads[1:])))
# one slow step: list every kernel between two heads whose delta > 18 ms
for a, b in zip(heads, heads[1:]):
    if (b[0] - a[0]) / 1e6 > 18:
        prev = a[1]
        for s, e, n in ev:
            if s <= a[0] or s > b[0]:
                continue
            gap = (s - prev) / 1e3
            if gap > 100 or (e - s) / 1e3 > 300:
                print("%10.1f us gap %8.1f dur %8.1f %s" % ((s - a[0]) / 1e3, gap, (e - s) / 1e3, n))
            prev = max(prev, e)
        break
