"""One compact line from bench.py's JSON (stdin): value, ms/step, per-kernel averages, generation counters.  Usage: python bench.py ... | python tools/ab_line.py TAG"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
pk = {k: round(v["avg_us"], 1) for k, v in (d.get("kernels") or d["roofline"])["per_kernel"].items()}
wr = d.get("workload_rates", {}).get("per_batch_step", {})
print(tag, round(d["value"] / 1e6, 1), "M", round(d["ms_per_step"] * 1e3, 1), "us", pk,
      {k: round(wr[k], 1) for k in ("descents", "inline_generations", "next_level_structures_used") if k in wr})
