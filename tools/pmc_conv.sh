#!/bin/bash
# PMC passes over one microbenchmarked conv layer (development aid).  usage: tools/pmc_conv.sh <layer> <pass> <outdir> [extra microbench args, e.g. --dtype fp32]
set -e
LAYER=${1:-c2c_3x1x1}; PASS=${2:-fwd}; OUT=${3:-gpurun_out/pmcconv}; EXTRA="${@:4}"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVES SQ_INSTS_SMEM" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python tools/conv_microbench.py --layers $LAYER --passes $PASS --reps 3 $EXTRA > $OUT.p$i.log 2>&1
  echo "pass $i done"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv_' not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in agg.items():
    print(k[:120])
    wc = d.get('SQ_WAVE_CYCLES', 1.0)
    for c in sorted(d):
        print('   %-32s %14.4g  %6.1f%% of WAVE_CYCLES' % (c, d[c], 100.0 * d[c] / wc))
PY
