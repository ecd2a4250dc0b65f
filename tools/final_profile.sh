#!/bin/bash
# Round-end measurement on the GPU box: bench line, rocprofv3 kernel statistics of the same command, the two PMC passes
# behind roofline.traffic, and the per-launch table -- for the headline leg (fp32, the reference's arithmetic) and for the
# bf16 secondary leg.  usage: tools/final_profile.sh <outdir under gpurun_out> <round tag>
# Afterwards copy <outdir>/<dtype>/{bench_prof.log,launches.tsv,pmc_traffic.json,stats/**/*kernel_stats.csv} and
# <outdir>/bench_full.log into profiles/<tag>_*.
set -e -o pipefail
OUT=${1:-gpurun_out/fin}; TAG=${2:-r04}
mkdir -p $OUT
export TMPDIR=/tmp
echo "[0] bench (default flags: fp32 headline + bf16 secondary + cpu baseline)"
timeout -k 10 500 python bench.py > $OUT/bench_full.log 2> $OUT/bench_full.err
for DT in fp32 bf16; do
  O=$OUT/$DT; mkdir -p $O
  ARGS="--dtype $DT --secondary none --no-cpu-baseline"
  echo "[$DT 1/4] rocprofv3 --kernel-trace --stats"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 10 --warmup 3 $ARGS > $O/bench_prof.log 2>&1
  echo "[$DT 2/4] PMC FETCH_SIZE"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python bench.py --steps 3 --warmup 2 $ARGS > $O/pmc_f.log 2>&1
  echo "[$DT 3/4] PMC WRITE_SIZE"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python bench.py --steps 3 --warmup 2 $ARGS > $O/pmc_w.log 2>&1
  python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json
  echo "[$DT 4/4] per-launch table"
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 $ARGS --dump-launches $O/launches.tsv > $O/bench_launch.log 2>&1
  find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  rm -rf $O/pmc_f $O/pmc_w $O/stats
done
grep -h "^{\"metric" $OUT/bench_full.log | tail -c 600
