#!/bin/bash
# Development aid: SQ / LDS / TA counters of the fp32 weight gradient on one layer, per DUALVAR_WGRAD_F32S configuration.
# usage: tools/pmc_wgrad.sh <outdir> <layer> <cfg ...>   (cfg "d" = default kernel)
OUT=$1; LAYER=$2; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
for c in "$@"; do
  if [ "$c" = "d" ]; then unset DUALVAR_WGRAD_F32S; else export DUALVAR_WGRAD_F32S=$c; fi
  tools/pmc_kernel.sh $OUT/$LAYER.$c conv_wgrad tools/conv_microbench.py --dtype fp32 --layers $LAYER --passes wgrad --reps 3 > $OUT/$LAYER.$c.txt 2>&1
  rm -rf $OUT/$LAYER.$c
done
