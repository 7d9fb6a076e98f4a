"""Copy what tools/collect_profiles.sh <tag> left in gpurun_out/ into profiles/ (kernel stats rendered as text) and print the figures the docs quote.
Usage: python tools/publish_profiles.py r02"""
import csv, json, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
for src, dst in (("gpurun_out/%s_kernel_stats.csv" % tag, "profiles/%s_kernel_stats.txt" % tag), ("gpurun_out/%s_kernel_stats_driver_cmd.csv" % tag, "profiles/%s_kernel_stats_driver_cmd.txt" % tag)):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write("rocprofv3 --kernel-trace --stats\n%-100s %8s %12s %10s %7s %9s %9s\n" % ("Name", "Calls", "TotalNs", "AvgNs", "Pct", "MinNs", "MaxNs"))
        for r in rows:
            f.write("%-100s %8s %12s %10.0f %7s %9s %9s\n" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
        print(dst, [(r["Name"][:12], round(float(r["AverageNs"]) / 1e3, 1)) for r in rows[:3]])
for n in ("bench.json", "bench_driver_cmd.json", "pmc_traffic.json", "pmc_calibration.json", "wave_profile.txt", "value_api.txt", "driver_repro_after.txt"):
    shutil.copy("gpurun_out/%s_%s" % (tag, n), "profiles/%s_%s" % (tag, n))
shutil.copy("gpurun_out/%s_pmc_traffic.json" % tag, "profiles/pmc_traffic.json")
for f in ("profiles/%s_bench.json" % tag, "profiles/%s_bench_driver_cmd.json" % tag):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f)
    print(" value %.1f M  ms %.4f  frac %.3f (%.0f GB/s)  e2e %.3f  copy peak %.0f GB/s  frac of copy peak %.3f  traffic %.2f MB" %
          (d["value"] / 1e6, d["ms_per_step"], r["frac"], r["achieved"], r["frac_end_to_end"], r["copy_peak_GBps"], r["frac_of_copy_peak"], r["traffic"] / 1e6))
    print(" per_kernel", {k: (round(v["avg_us"], 1), v["launches"], round(v.get("frac_of_copy_peak", 0), 3), round(v["algo_GBps"])) for k, v in r["per_kernel"].items()})
    print(" repeats", [round(x * 1e3, 1) for x in d["repeats"]["ms_per_step"]], "median %.1f M" % (d["repeats"]["median_value"] / 1e6))
    print(" cold start %.1f M" % (d["preroll"]["cold_start"]["value"] / 1e6))
    print(" cpu %.2f M" % (d["cpu_baseline"]["value"] / 1e6), {k: round(v["value"] / 1e6, 2) for k, v in d["cpu_baseline"]["other_sizes"].items()})
    print(" per step", {k: round(v, 1) for k, v in d["workload_rates"]["per_batch_step"].items()})
    for k, w in d.get("extra_workloads", {}).items():
        print(" extra", k, "%.1f M" % (w["value"] / 1e6), round(w["ms_per_step"], 4), {kk: round(v["avg_us"], 1) for kk, v in w["per_kernel"].items()}, "k_obs %.0f GB/s" % w["per_kernel"]["k_obs"]["algo_GBps"])
t = json.load(open("profiles/%s_pmc_traffic.json" % tag))
print({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in t.items() if k[0] == "k"})
