#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel statistics of the same command, the two PMC passes
# behind roofline.traffic, and the per-launch table.  usage: tools/final_profile.sh <outdir under gpurun_out> <round tag>
# Afterwards copy <outdir>/{bench_full.log,bench_prof.log,launches.tsv,pmc_traffic.json,stats/**/*kernel_stats.csv}
# into profiles/<tag>_final_*.
set -e -o pipefail
OUT=${1:-gpurun_out/fin}; TAG=${2:-r01}
mkdir -p $OUT
export TMPDIR=/tmp
echo "[1/5] bench (default flags)"
timeout -k 10 400 python bench.py > $OUT/bench_full.log 2> $OUT/bench_full.err
echo "[2/5] rocprofv3 --kernel-trace --stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
echo "[3/5] PMC FETCH_SIZE"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_f.log 2>&1
echo "[4/5] PMC WRITE_SIZE"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_w.log 2>&1
python tools/pmc_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/pmc_traffic.json
echo "[5/5] per-launch table"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dump-launches $OUT/launches.tsv > $OUT/bench_launch.log 2>&1
# counter CSVs are large: keep the summaries only
rm -rf $OUT/pmc_f/*/*counter_collection.csv.bak
tail -c 400 $OUT/bench_full.log
