#!/bin/bash
# Development aid: the headline bench under several environment settings, one line per setting.
# usage: tools/ab_bench.sh <outdir> "VAR=val VAR2=val" "VAR=val" ...      ("-" = no override)
OUT=$1; shift
mkdir -p $OUT
i=0
for setting in "$@"; do
  i=$((i+1))
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --secondary none ${BENCH_ARGS} > $OUT/ab$i.log 2> $OUT/ab$i.err
  python - "$OUT/ab$i.log" "$setting" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    k = d['kernel_time_ms_per_step']
    print('%-40s %9.1f clips/s %7.3f ms loss %.4f | %s' % (sys.argv[2], d['value'], d['ms_per_step'], d['loss'],
          ' '.join('%s=%.2f' % (a.replace('conv_', '').replace('<f32,', '<'), b) for a, b in list(k.items())[:9])))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
