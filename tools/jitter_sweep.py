"""Step-latency jitter against the level-per-lane producer's launch shape (VERDICT r5 task 5b): for each setting of the development library's knobs, one
`bench.py --steps 20 --warmup 5` run; prints its long-window rate and the per-step GPU-time percentiles (bench.py `step_us`: an event pair on every launch of a
1 000-step window).  Usage: python tools/jitter_sweep.py [mini|default]  (needs rogue-gym_amd/variants/librogue_gym_hip_dev.so: __graft_entry__.build())."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "rogue-gym_amd", "variants", "librogue_gym_hip_dev.so")
wl = sys.argv[1] if len(sys.argv) > 1 else "mini"
SETS = [("product library", None),
        ("waves 256 every 16 (= product)", {}),
        ("waves 128 every 16", {"ROGUE_GYM_HIP_LANE_WAVES": "128"}),
        ("waves 64 every 16", {"ROGUE_GYM_HIP_LANE_WAVES": "64"}),
        ("waves 32 every 16", {"ROGUE_GYM_HIP_LANE_WAVES": "32"}),
        ("waves 16 every 16", {"ROGUE_GYM_HIP_LANE_WAVES": "16"}),
        ("waves 64 every 8", {"ROGUE_GYM_HIP_LANE_WAVES": "64", "ROGUE_GYM_HIP_LANE_EVERY": "8"}),
        ("waves 32 every 8", {"ROGUE_GYM_HIP_LANE_WAVES": "32", "ROGUE_GYM_HIP_LANE_EVERY": "8"}),
        ("waves 32 every 4", {"ROGUE_GYM_HIP_LANE_WAVES": "32", "ROGUE_GYM_HIP_LANE_EVERY": "4"}),
        ("waves 256 every 16, 32 levels per wave", {"ROGUE_GYM_HIP_LANE_L": "32"}),
        ("waves 256 every 16, 16 levels per wave", {"ROGUE_GYM_HIP_LANE_L": "16"}),
        ("waves 256 every 32", {"ROGUE_GYM_HIP_LANE_EVERY": "32"}),
        ("spares kept: no producer (floor)", {"ROGUE_GYM_HIP_KEEP_SPARES": "1"})]
print("# workload %s: value_long_window M env-steps/s | step us p50 p90 p99 max | k_step us p50 p90 p99 max" % wl)
for name, knobs in SETS:
    env = dict(os.environ)
    if knobs is not None:
        env.update(knobs, ROGUE_GYM_HIP_LIB=DEV)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extra", "--no-repeats", "--workload", wl]
    if wl != "mini":
        cmd += ["--preroll-steps", "500"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT)
    try:
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        s, k = d["step_us"]["step"], d["step_us"]["k_step"]
        print("%-44s %7.1f | %6.1f %6.1f %6.1f %6.1f | %6.1f %6.1f %6.1f %6.1f" % (name, d["value_long_window"] / 1e6, s["p50"], s["p90"], s["p99"], s["max"], k["p50"], k["p90"], k["p99"], k["max"]), flush=True)
    except Exception as e:  # noqa: BLE001
        print(name, "FAILED", e, r.stderr[-300:], flush=True)
