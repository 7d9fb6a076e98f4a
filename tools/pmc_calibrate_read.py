import csv, glob, json, sys
out = {}
for d, counter in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    # the two big copy kernels of tools/pmc_calibrate.py, in dispatch order: wide (1 GiB in, 1 GiB out), narrow (2^20 lines in, 2 MiB out)
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    big = [r for r in rows if float(r["Counter_Value"]) > 1000][-2:]
    out[counter] = [{"kernel": r["Kernel_Name"][:60], "value_KiB": float(r["Counter_Value"])} for r in big]
w, nrw = out["FETCH_SIZE"]
out["wide_fetch_bytes_counted_per_byte_read"] = w["value_KiB"] * 1024 / float(1 << 30)
out["narrow_fetch_bytes_counted_per_line_touched"] = nrw["value_KiB"] * 1024 / float(1 << 20)
ww, nw = out["WRITE_SIZE"]
out["wide_write_bytes_counted_per_byte_written"] = ww["value_KiB"] * 1024 / float(1 << 30)
out["narrow_write_bytes_counted_per_2B_element"] = nw["value_KiB"] * 1024 / float(1 << 20)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
