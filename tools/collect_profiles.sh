#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the repo root):  tools/collect_profiles.sh r02
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-extra --no-repeats"
# 1. the default bench line (everything: extras, repeats, cpu baseline) and the DRIVER's exact command
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2>> $out/${tag}_bench.err
# 2. rocprofv3 kernel stats: of the driver's command and of the 2000-step run
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_trace_drv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $Q --clock-warm-s 0 > $out/${tag}_trace_drv.log 2>&1
cp "$(find /tmp/${tag}_trace_drv -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_driver_cmd.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_trace -- python bench.py --steps 2000 --warmup 200 $Q --clock-warm-s 0 > $out/${tag}_trace.log 2>&1
cp "$(find /tmp/${tag}_trace -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats.csv
# 3. HBM traffic: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), every launch on 65 536 envs
P="--steps 60 --warmup 20 --preroll-steps 200 --clock-warm-s 0 $Q"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${tag}_fetch -- python bench.py $P > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${tag}_write -- python bench.py $P > $out/${tag}_write.log 2>&1
fd=$(dirname $(find /tmp/${tag}_fetch -name "*counter_collection.csv" | head -1))
wd=$(dirname $(find /tmp/${tag}_write -name "*counter_collection.csv" | head -1))
# 4. calibration of the two counters on this box for the stepper's two access patterns
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${tag}_cal_f -- python tools/pmc_calibrate.py > $out/${tag}_cal.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${tag}_cal_w -- python tools/pmc_calibrate.py >> $out/${tag}_cal.log 2>&1
python tools/pmc_calibrate_read.py /tmp/${tag}_cal_f /tmp/${tag}_cal_w $out/${tag}_pmc_calibration.json > /dev/null 2>> $out/${tag}_cal.log
python tools/pmc_summary.py "$fd" "$wd" $out/${tag}_pmc_traffic.json $out/${tag}_pmc_calibration.json > /dev/null 2>> $out/${tag}_cal.log
# 5. per-wave phase profile, value-object API rates, the driver command A/B
python tools/microbench.py prof1 2>&1 | grep -v amdgpu | cut -c1-400 > $out/${tag}_wave_profile.txt
python tools/bench_value_api.py 64 1024 8192 65536 2>&1 | grep -v amdgpu > $out/${tag}_value_api.txt
bash tools/driver_repro.sh 2>&1 | tail -9 > $out/${tag}_driver_repro_after.txt
cat $out/${tag}_bench_driver_cmd.json | head -c 600
