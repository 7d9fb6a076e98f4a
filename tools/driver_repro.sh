#!/bin/bash
# Evidence for the driver-run bench line: the driver's exact command under the four combinations of
# {first spares awaited in rg_create | left in the background (round-1 behaviour)} x {clock-warm phase on | off}.
# Writes gpurun_out/driver_repro_*.json (one bench line each).
mkdir -p gpurun_out
for spares in wait async; do
  for warm in 1.5 0; do
    if [ $spares = async ]; then export ROGUE_GYM_HIP_ASYNC_FIRST_SPARES=1; else unset ROGUE_GYM_HIP_ASYNC_FIRST_SPARES; fi
    for rep in 1 2; do
      python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-repeats --clock-warm-s $warm > gpurun_out/driver_repro_${spares}_warm${warm}_$rep.json 2> gpurun_out/driver_repro.err
    done
  done
done
unset ROGUE_GYM_HIP_ASYNC_FIRST_SPARES
python3 - <<'P'
import glob, json
for f in sorted(glob.glob("gpurun_out/driver_repro_*_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        pk = d["roofline"]["per_kernel"]
        print("%-52s %7.1f M/s  %.3f ms  k_step %6.1f us  k_obs %5.1f us  sclk %s -> %s" % (f, d["value"] / 1e6, d["ms_per_step"], pk["k_step"]["avg_us"], pk["k_obs"]["avg_us"],
              (d.get("clock_warm") or {}).get("sclk_mhz_before"), d.get("sclk_mhz_after_timed_region")))
    except Exception as e:
        print(f, "ERR", e)
P
