#!/bin/bash
# One round's evidence, one gpurun call on one box:  tools/collect_profiles.sh <tag, e.g. r05> <commit>
# For EACH workload of SURVEY.md 8(d) (mini = the headline, default 80x24, nohide-symbol): rocprofv3 --kernel-trace --stats, calibrated PMC traffic
# (separate FETCH_SIZE / WRITE_SIZE passes) and SQ counters (two passes), so that per-kernel roofline fractions of all three are recomputable
# from profiles/ alone.  Plus the bench lines, the wave profile and the value-object API rates.
set -u
tag=${1:-r04}; commit=${2:-unknown}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
Q="--no-cpu-baseline --no-extra --no-repeats"
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# 1. bench lines: default run (extras, repeats, cpu baseline) and the DRIVER's exact command
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_driver_cmd.json 2>> $out/${tag}_bench.err
# 2. calibration of FETCH_SIZE / WRITE_SIZE on this box
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${tag}_cal_f -- python tools/pmc_calibrate.py > $out/${tag}_cal.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${tag}_cal_w -- python tools/pmc_calibrate.py >> $out/${tag}_cal.log 2>&1
python tools/pmc_calibrate_read.py /tmp/${tag}_cal_f /tmp/${tag}_cal_w $out/${tag}_pmc_calibration.json > /dev/null 2>> $out/${tag}_cal.log
# 3. per workload: kernel stats, PMC traffic, SQ counters
for wl in mini default nohide-symbol; do
  case $wl in
    mini) S="--steps 2000 --warmup 200"; P="--steps 60 --warmup 20 --preroll-steps 200";;
    default) S="--steps 400 --warmup 50 --preroll-steps 500"; P="--steps 40 --warmup 10 --preroll-steps 200";;
    *) S="--steps 80 --warmup 10 --preroll-steps 100"; P="--steps 20 --warmup 5 --preroll-steps 60";;
  esac
  W="--workload $wl $Q --clock-warm-s 0"
  python bench.py --workload $wl --no-extra --no-cpu-baseline $S > $out/${tag}_bench_${wl}.json 2>> $out/${tag}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_trace_$wl -- python bench.py $W $S > $out/${tag}_trace_$wl.log 2>&1
  cp "$(find /tmp/${tag}_trace_$wl -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_$wl.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/${tag}_fetch_$wl -- python bench.py $W $P > $out/${tag}_pmc_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/${tag}_write_$wl -- python bench.py $W $P >> $out/${tag}_pmc_$wl.log 2>&1
  fd=$(dirname $(find /tmp/${tag}_fetch_$wl -name "*counter_collection.csv" | head -1))
  wd=$(dirname $(find /tmp/${tag}_write_$wl -name "*counter_collection.csv" | head -1))
  python tools/pmc_summary.py "$fd" "$wd" $out/${tag}_pmc_traffic_$wl.json $out/${tag}_pmc_calibration.json > /dev/null 2>> $out/${tag}_cal.log
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d /tmp/${tag}_sq1_$wl -- python bench.py $W $P >> $out/${tag}_pmc_$wl.log 2>&1
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d /tmp/${tag}_sq2_$wl -- python bench.py $W $P >> $out/${tag}_pmc_$wl.log 2>&1
  { echo "# workload $wl, commit $commit: rocprofv3 --kernel-trace --pmc <SQ counters> -- python bench.py $W $P (two passes); averages per launch (tools/pmc_sq.py)";
    python tools/pmc_sq.py "$(dirname $(find /tmp/${tag}_sq1_$wl -name '*counter_collection.csv' | head -1))";
    python tools/pmc_sq.py "$(dirname $(find /tmp/${tag}_sq2_$wl -name '*counter_collection.csv' | head -1))"; } > $out/${tag}_sq_counters_$wl.txt 2>> $out/${tag}_cal.log
done
# the driver's command under rocprofv3 too (mini)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_trace_drv -- python3 bench.py --gpus 1 --steps 20 --warmup 5 $Q --clock-warm-s 0 > $out/${tag}_trace_drv.log 2>&1
cp "$(find /tmp/${tag}_trace_drv -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_driver_cmd.csv
# 4. per-wave phase profiles (mini and default), value-object API
python tools/microbench.py prof1 2>&1 | grep -v amdgpu | cut -c1-400 > $out/${tag}_wave_profile.txt
python tools/microbench.py profd 2>&1 | grep -v amdgpu | cut -c1-400 > $out/${tag}_wave_profile_default.txt
python tools/bench_value_api.py 64 1024 8192 65536 2>&1 | grep -v amdgpu > $out/${tag}_value_api.txt
# 4b. round 6: the bound observation tensor (opt-in, never the headline): bench lines of the three workloads, kernel stats of the mini run, the bound step kernel's wave profile;
#     step-latency percentiles against the producer's launch shape (development library)
for wl in mini default nohide-symbol; do
  case $wl in
    mini) S="--steps 2000 --warmup 200";;
    default) S="--steps 400 --warmup 50 --preroll-steps 500";;
    *) S="--steps 80 --warmup 10 --preroll-steps 100";;
  esac
  python bench.py --workload $wl --no-extra --no-cpu-baseline --no-repeats --persistent-obs $S > $out/${tag}_bench_bound_${wl}.json 2>> $out/${tag}_bench.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/${tag}_trace_bound -- python bench.py --workload mini $Q --clock-warm-s 0 --persistent-obs --steps 2000 --warmup 200 > $out/${tag}_trace_bound.log 2>&1
cp "$(find /tmp/${tag}_trace_bound -name '*kernel_stats.csv' | head -1)" $out/${tag}_kernel_stats_bound_mini.csv
RG_PROF_BOUND=1 python tools/microbench.py prof1 2>&1 | grep -v amdgpu | cut -c1-400 > $out/${tag}_wave_profile_bound.txt
python tools/jitter_sweep.py mini > $out/${tag}_jitter_mini.txt 2>&1
python tools/jitter_sweep.py default > $out/${tag}_jitter_default.txt 2>&1
# 5. ONE fuzz soak (random valid configs, HIP vs oracle in lock step)
python tools/fuzz_parity.py --minutes 4 --seed 404 2>&1 | grep -v amdgpu | tail -40 > $out/${tag}_fuzz_parity.txt
echo "$commit" > $out/${tag}_commit.txt
python -c "import __graft_entry__ as g; print(g.library_id(g.PKG + '/librogue_gym_hip.so'))" > $out/${tag}_build_id.txt   # the library the PMC passes ran on (bench.py withholds roofline.traffic on any other build)
head -c 700 $out/${tag}_bench_driver_cmd.json
