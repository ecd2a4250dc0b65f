#!/bin/bash
# Development aid: correctness (op tests) and speed (tools/conv_microbench.py, wgrad pass) of every conv_wgrad_f32s_kernel
# configuration against the default fp32 weight gradient.  usage: tools/wgrad_sweep.sh <outdir> [configs...]   (0..5)
OUT=${1:-gpurun_out/wg}; shift
CFGS=${@:-"0 1 2 3 4 5"}
mkdir -p $OUT
echo "== default" | tee $OUT/sweep.log
timeout -k 10 120 python tools/conv_microbench.py --dtype fp32 --layers all --passes wgrad >> $OUT/sweep.log 2>&1
for c in $CFGS; do
  echo "== DUALVAR_WGRAD_F32S=$c" | tee -a $OUT/sweep.log
  DUALVAR_WGRAD_F32S=$c timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad or conv_fwd_dgrad" > $OUT/test_$c.log 2>&1
  echo "tests rc=$? $(tail -1 $OUT/test_$c.log)" | tee -a $OUT/sweep.log
  DUALVAR_WGRAD_F32S=$c timeout -k 10 120 python tools/conv_microbench.py --dtype fp32 --layers all --passes wgrad >> $OUT/sweep.log 2>&1
done
