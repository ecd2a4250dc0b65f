#!/bin/bash
# The other workloads of DESIGN.md section 6 (one GPU): every run reports the fp32 headline leg and the bf16 secondary leg.
# usage: tools/bench_matrix.sh <outfile>
OUT=${1:-gpurun_out/matrix.jsonl}; : > $OUT
run() { echo "== $*"; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $OUT || echo "{\"failed\": \"$*\"}" >> $OUT; }
run --steps 20 --warmup 5
run --steps 12 --warmup 4 --batch 128
run --steps 12 --warmup 4 --frames 16
run --steps 12 --warmup 4 --model moco_naked
run --steps 10 --warmup 3 --model moco_timeseriesv4 --batch 32
run --steps 10 --warmup 3 --model simclr_timeseriesv4 --batch 32
run --steps 8 --warmup 3 --model simclr_timeseriesv4 --net r21d --batch 32
run --steps 12 --warmup 4 --net r3d --batch 32
run --steps 6 --warmup 3 --net r50 --frames 32 --size 224 --batch 4
run --steps 6 --warmup 3 --net r50 --frames 32 --size 224 --batch 4 --dtype fp8pw --secondary none
DUALVAR_F32_EXACT=1 run --steps 10 --warmup 3 --secondary none
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if 'failed' in d: print('FAILED', d['failed']); continue
    s = d.get('secondary')
    print('%-96s %5s %9.1f clips/s %8.2f ms   %s' % (d['config']['workload'][:96], d['dtype'], d['value'], d['ms_per_step'],
          ('| %s %9.1f clips/s %8.2f ms' % (s['dtype'], s['value'], s['ms_per_step'])) if s else ''))
PY
