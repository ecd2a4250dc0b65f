#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over the pool microbenchmark with a known-byte copy beside it (calibration of the
# counters and of the x2 gfx950 FETCH_SIZE correction).  usage: tools/pmc_pool.sh <outdir> [pool_microbench args]
set -e
OUT=${1:-gpurun_out/pmcpool}; EXTRA="${@:2}"
export TMPDIR=/tmp
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag -- python tools/pool_microbench.py --calib --reps 2 $EXTRA > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not any(s in k for s in ('pool', 'opy', 'elementwise')): continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, d in sorted(agg.items()):
    print(k[:110])
    for c in sorted(d):
        print('   %-28s per launch %14.1f   (launches %d)' % (c, d[c] / cnt[k][c], cnt[k][c]))
PY
