#!/bin/bash
# SQ / TCC / traffic PMC passes over any microbenchmark command (development aid).
# usage: tools/pmc_kernel.sh <outdir> <kernel name filter> <python script> [args ...]
set -e
OUT=$1; FILT=$2; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INSTS_SMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - "$OUT" "$FILT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if sys.argv[2] not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, d in sorted(agg.items()):
    print(k[:120])
    wc = d.get('SQ_WAVE_CYCLES', 1.0) / max(cnt[k]['SQ_WAVE_CYCLES'], 1)
    for c in sorted(d):
        v = d[c] / cnt[k][c]
        print('   %-32s %14.5g per launch  %6.1f%% of WAVE_CYCLES' % (c, v, 100.0 * v / wc))
PY
