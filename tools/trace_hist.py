"""Histogram of per-dispatch kernel durations from a rocprofv3 --kernel-trace CSV (development aid)."""
import csv, glob, sys, collections
import numpy as np
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    k = "k_step" if k.startswith("k_step") else k
    d[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k in ("k_step", "k_obs<0>", "k_regen"):
    if k not in d:
        continue
    v = sorted(d[k])
    dur = np.array([(b - a) / 1e3 for a, b in v])
    print(k, "n=%d mean %.1f median %.1f p10 %.1f p90 %.1f max %.1f" % (len(dur), dur.mean(), np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90), dur.max()))
    print("   hist 20us buckets:", np.histogram(dur, bins=np.arange(0, 420, 20))[0].tolist())
    if k == "k_step":
        print("   first 40:", " ".join("%.0f" % x for x in dur[:40]))
        print("   last 40:", " ".join("%.0f" % x for x in dur[-40:]))
        big = np.where(dur > 250)[0]
        print("   >250us at launch idx:", big.tolist()[:60])
