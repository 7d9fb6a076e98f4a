Q="--no-cpu-baseline --no-extra --no-repeats --steps 600 --warmup 100"
for v in head new r32 head new r32; do
  if [ $v = new ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  python bench.py $Q | python tools/ab_line.py $v
done
unset ROGUE_GYM_HIP_LIB
python -m pytest tests/test_gpu_features.py -x -q -k "forty" 2>&1 | tail -3
