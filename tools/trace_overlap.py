import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
names = [r["Kernel_Name"].split("(")[0] for r in rows]
idx = [i for i, nme in enumerate(names) if nme == "k_regen"]
for i in idx[:6]:
    for j in range(max(0, i - 3), min(len(rows), i + 6)):
        r = rows[j]
        print("%s%-10s q=%s start %.1f us dur %.1f us" % ("* " if j == i else "  ", names[j][:10], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print()
