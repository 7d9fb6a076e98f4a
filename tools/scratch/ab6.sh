Q="--no-cpu-baseline --no-extra --no-repeats --steps 600 --warmup 100"
for v in head new head new; do
  if [ $v = new ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  python bench.py $Q | python tools/ab_line.py $v
done
for v in head new; do
  if [ $v = new ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  python bench.py --workload default --no-cpu-baseline --no-extra --no-repeats --steps 300 --warmup 50 --preroll-steps 500 | python tools/ab_line.py default-$v
done
unset ROGUE_GYM_HIP_LIB
python tools/microbench.py gen1 2>&1 | grep k_build | head -3
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
