"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json (HBM bytes per launch per kernel).
FETCH_SIZE / WRITE_SIZE are reported in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE counts exactly half of a WIDE coalesced streaming
read (16 B per lane) and is uncalibrated for other widths -- so the factor per kernel comes from a calibration on this box
(tools/pmc_calibrate.py): k_obs / k_regen use the WIDE factor, k_step (one 2-byte word per lane, a different env's grid per lane) the NARROW
one, where a touched line is counted as the 64 bytes that are actually fetched (factor 1)."""
import csv, glob, json, sys, collections

def load(d, counter):
    f = glob.glob(d + "/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_step"):
            k = "k_step"  # (k_step_w32 = the W <= 32 instance, capped at two waves per SIMD)
        acc[k][0] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    return {k: (v[0], len(seen[k])) for k, v in acc.items()}

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
cal = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
wide = 1.0 / cal["wide_fetch_bytes_counted_per_byte_read"] if cal.get("wide_fetch_bytes_counted_per_byte_read") else 2.0
wide_w = 1.0 / cal["wide_write_bytes_counted_per_byte_written"] if cal.get("wide_write_bytes_counted_per_byte_written") else 1.0
out = {}
for k in ("k_step", "k_obs", "k_regen", "k_classify"):
    if k in fetch and k in write:
        f = fetch[k][0] / fetch[k][1] * 1024.0
        w = write[k][0] / write[k][1] * 1024.0
        ff = 1.0 if k in ("k_step", "k_classify") else wide  # narrow per-lane accesses: a request is counted as the 64-byte line it fetches
        out[k] = {"fetch_bytes_raw_per_launch": f, "fetch_factor": ff, "fetch_bytes_per_launch": ff * f, "write_bytes_raw_per_launch": w,
                  "write_factor": wide_w, "write_bytes_per_launch": wide_w * w, "hbm_bytes_per_launch": ff * f + wide_w * w, "launches": fetch[k][1]}
out["_calibration"] = cal
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
