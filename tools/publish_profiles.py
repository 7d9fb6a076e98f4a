"""Copy what tools/collect_profiles.sh <tag> <commit> left in gpurun_out/ into profiles/ (kernel stats as text, PMC traffic stamped with the commit
it was measured on) and print per-kernel roofline fractions of every workload.  Usage: python tools/publish_profiles.py r04"""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
commit = open("gpurun_out/%s_commit.txt" % tag).read().strip()
build_id = open("gpurun_out/%s_build_id.txt" % tag).read().strip()
ALGO = {"mini": (65536, {"k_step": 128, "k_obs": 3072}, 3200), "default": (32768, {"k_step": 256, "k_obs": 11520}, 11776), "nohide-symbol": (32768, {"k_step": 256, "k_obs": 334080}, 334336)}
def stats_txt(src, dst):
    rows = list(csv.DictReader(open(src)))
    with open(dst, "w") as f:
        f.write("rocprofv3 --kernel-trace --stats   (commit %s)\n%-100s %8s %12s %10s %7s %9s %9s\n" % (commit, "Name", "Calls", "TotalNs", "AvgNs", "Pct", "MinNs", "MaxNs"))
        for r in rows:
            f.write("%-100s %8s %12s %10.0f %7s %9s %9s\n" % (r["Name"][:100], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
    return {r["Name"].split("(")[0].replace("void ", "").split("<")[0]: float(r["AverageNs"]) / 1e3 for r in rows}
summary = {"commit": commit, "peak_GBps": 8000, "workloads": {}}
for wl, (n, per_kernel, total) in ALGO.items():
    avg = stats_txt("gpurun_out/%s_kernel_stats_%s.csv" % (tag, wl), "profiles/%s_kernel_stats_%s.txt" % (tag, wl))
    t = json.load(open("gpurun_out/%s_pmc_traffic_%s.json" % (tag, wl)))
    t["_build_id"] = build_id
    t["_measured"] = "commit %s, workload %s, tools/collect_profiles.sh %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated on the same box)" % (commit, wl, tag)
    json.dump(t, open("profiles/%s_pmc_traffic_%s.json" % (tag, wl), "w"), indent=1)
    shutil.copy("gpurun_out/%s_sq_counters_%s.txt" % (tag, wl), "profiles/%s_sq_counters_%s.txt" % (tag, wl))
    shutil.copy("gpurun_out/%s_bench_%s.json" % (tag, wl), "profiles/%s_bench_%s.json" % (tag, wl))
    w = {}
    step_us = next((v for k, v in avg.items() if k.startswith("k_step")), None)
    obs_us = avg.get("k_obs")
    for k, us in (("k_step", step_us), ("k_obs", obs_us)):
        if us:
            gb = per_kernel[k] * n / (us * 1e-6) / 1e9
            tr = t.get(k, {}).get("hbm_bytes_per_launch")
            w[k] = {"avg_us_rocprof": round(us, 2), "algo_bytes_per_launch": per_kernel[k] * n, "algo_GBps": round(gb, 1), "frac_of_peak": round(gb / 8000, 4),
                    "pmc_traffic_bytes_per_launch": tr, "traffic_over_algorithmic": round(tr / (per_kernel[k] * n), 3) if tr else None}
    if step_us:
        w["contract_frac_dominant_kernel"] = round(total * n / (max(step_us, obs_us or 0) * 1e-6) / 1e9 / 8000, 4)
    summary["workloads"][wl] = w
    print(wl, json.dumps(w))
json.dump(summary, open("profiles/%s_roofline_summary.json" % tag, "w"), indent=1)
stats_txt("gpurun_out/%s_kernel_stats_driver_cmd.csv" % tag, "profiles/%s_kernel_stats_driver_cmd.txt" % tag)
shutil.copy("profiles/%s_kernel_stats_mini.txt" % tag, "profiles/%s_kernel_stats.txt" % tag)
if os.path.exists("gpurun_out/%s_kernel_stats_bound_mini.csv" % tag):
    stats_txt("gpurun_out/%s_kernel_stats_bound_mini.csv" % tag, "profiles/%s_kernel_stats_bound_mini.txt" % tag)
for n in ("bench.json", "bench_driver_cmd.json", "pmc_calibration.json", "wave_profile.txt", "wave_profile_default.txt", "value_api.txt", "fuzz_parity.txt",
          "bench_bound_mini.json", "bench_bound_default.json", "bench_bound_nohide-symbol.json", "wave_profile_bound.txt", "jitter_mini.txt", "jitter_default.txt", "eight_ranks_wall.txt"):
    if not os.path.exists("gpurun_out/%s_%s" % (tag, n)):
        continue
    shutil.copy("gpurun_out/%s_%s" % (tag, n), "profiles/%s_%s" % (tag, n))
shutil.copy("profiles/%s_pmc_traffic_mini.json" % tag, "profiles/pmc_traffic.json")
for f in ("profiles/%s_bench.json" % tag, "profiles/%s_bench_driver_cmd.json" % tag):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "value %.1f M  ms %.4f  frac %.3f  e2e %.3f  median %.1f M  cold %.1f M" % (d["value"] / 1e6, d["ms_per_step"], r["frac"], r["frac_end_to_end"],
          (d.get("value_median_of_repeats") or 0) / 1e6, (d.get("value_cold_start") or 0) / 1e6))
    print("  per_kernel", {k: round(v["avg_us"], 1) for k, v in d["kernels"]["per_kernel"].items()}, "cpu", round(d.get("cpu_baseline", {}).get("value", 0) / 1e6, 2), "long", round((d.get("value_long_window") or 0) / 1e6, 1))
    for k, w in d.get("extra_workloads", {}).items():
        print("  extra", k, "%.1f M" % (w["value"] / 1e6), {kk: round(v["avg_us"], 1) for kk, v in w["per_kernel"].items()})
