Q="--no-cpu-baseline --no-extra --no-repeats --steps 1000 --warmup 100"
python bench.py $Q | python tools/ab_line.py base
python bench.py $Q --time-every 101 | python tools/ab_line.py te101
ROGUE_GYM_HIP_REGEN_EVERY=4 python bench.py $Q | python tools/ab_line.py regen4
ROGUE_GYM_HIP_STEP_MARKER=1 python bench.py $Q | python tools/ab_line.py marker
python bench.py $Q | python tools/ab_line.py base
