#!/bin/bash
# End-to-end check of the entry script at the headline config, next to bench.py on the same box, and the rocprofv3 kernel
# statistics of the entry script.  usage: tools/e2e_profile.sh <outdir under gpurun_out>
set -e -o pipefail
OUT=${1:-gpurun_out/e2e}; mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--dataset synthetic-frames --net s3dg --model simclr_naked --batch_size 64 --seq_len 8 --num_seq 2 --img_dim 112 -j 4 --gpu 0 --epochs 1 --lr 0.003 --prefix $OUT/run"
timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --secondary none > $OUT/bench.log 2> $OUT/bench.err
timeout -k 10 400 python pretrain.py $ARGS --epoch_size 14080 --steps 220 --warm_steps 20 > $OUT/pretrain.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python pretrain.py $ARGS --epoch_size 5120 --steps 80 --warm_steps 20 > $OUT/pretrain_prof.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/stats $OUT/run*
grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log; grep -i -E "clips/s|steady" $OUT/pretrain.log | tail -3
