import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    k = "k_step" if k.startswith("k_step") else k
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in ("k_step", "k_obs", "k_regen"):
    if k in acc:
        n = len(cnt[k])
        print(k, "launches", n, {c: round(v / n) for c, v in acc[k].items()})
