python -m pytest tests/test_gpu_features.py -x -q -k "mixed_screen or init_items" 2>&1 | tail -5
Q="--no-cpu-baseline --no-extra --no-repeats --steps 600 --warmup 100"
for v in base rcp base rcp; do
  if [ $v = base ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  python bench.py $Q | python tools/ab_line.py $v
done
export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_rcp.so
python tools/microbench.py gen1 2>&1 | grep k_build
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
