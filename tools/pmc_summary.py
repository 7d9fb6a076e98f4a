"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json (HBM bytes per launch per kernel).
FETCH_SIZE / WRITE_SIZE are reported in KiB... units per MI355X_MICROARCH.md: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024,
and on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (correction applied: fetch x2)."""
import csv, glob, json, sys, collections

def load(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k][0] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    return {k: (v[0], len(seen[k])) for k, v in acc.items()}

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in ("k_step", "k_obs", "k_regen"):
    if k in fetch and k in write:
        f = fetch[k][0] / fetch[k][1] * 1024.0
        w = write[k][0] / write[k][1] * 1024.0
        out[k] = {"fetch_bytes_raw_per_launch": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes_per_launch": w,
                  "hbm_bytes_per_launch": 2 * f + w, "launches": fetch[k][1]}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
