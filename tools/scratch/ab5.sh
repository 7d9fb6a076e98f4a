for v in head new; do
  if [ $v = new ]; then unset ROGUE_GYM_HIP_LIB; else export ROGUE_GYM_HIP_LIB=$PWD/rogue-gym_amd/variants/librogue_$v.so; fi
  echo "== $v"; python tools/microbench.py prof1 2>&1 | grep -v amdgpu | cut -c1-330 | head -32
done
