#!/bin/bash
# Collect the per-round evidence on the GPU box (run through gpurun from the repo root):  tools/collect_profiles.sh v5
# Writes gpurun_out/<tag>_*; copy what should be judged into profiles/.
set -u
tag=${1:-vX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python bench.py --workload default --steps 500 --warmup 100 > $out/${tag}_bench_default.json 2>> $out/${tag}_bench.err
python bench.py --workload nohide-symbol --steps 100 --warmup 20 > $out/${tag}_bench_symbol.json 2>> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python bench.py --steps 60 --warmup 20 --no-cpu-baseline > $out/${tag}_write.log 2>&1
stats=$(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1)
cp "$stats" $out/${tag}_kernel_stats.csv
fd=$(dirname $(find $out/${tag}_fetch -name "*counter_collection.csv" | head -1))
wd=$(dirname $(find $out/${tag}_write -name "*counter_collection.csv" | head -1))
python tools/pmc_summary.py "$fd" "$wd" $out/${tag}_pmc_traffic.json > /dev/null
# keep the merged output small: drop the raw traces
rm -rf $out/${tag}_trace $out/${tag}_fetch $out/${tag}_write
cat $out/${tag}_bench.json
