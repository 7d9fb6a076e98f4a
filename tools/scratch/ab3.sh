python -m pytest tests/test_gpu_features.py -x -q -k "forty" 2>&1 | grep -v "^$" | tail -40 | cut -c1-300
Q="--no-cpu-baseline --no-extra --no-repeats --steps 600 --warmup 100"
for v in head new nocap head new nocap; do
  if [ $v = new ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  python bench.py $Q | python tools/ab_line.py $v
done
