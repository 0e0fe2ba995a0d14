#!/bin/bash
# SQ counters of the 64x64-level attention launch (one --pmc pass per counter group; no trace domains beside them)
set -u
OUT=gpurun_out/pmc_attn${MOBI_ATTN_V3:+_v3_$MOBI_ATTN_V3}      # (export MOBI_ATTN_V3=0 first for the kernel of rounds 1-2)
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python tools/kbench.py attn --heads 8 --dh 40 --t 4096 --images 16 --v-rows --iters 3 > $OUT/run$i.log 2> $OUT/err$i.log || { echo "pass $i failed"; tail -3 $OUT/err$i.log; }
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
import os
out = 'gpurun_out/pmc_attn' + ('_v3_' + os.environ['MOBI_ATTN_V3'] if os.environ.get('MOBI_ATTN_V3') else '')
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'attention_' not in r['Kernel_Name']: continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print(f"{k:32s} {v / n:16.0f} per launch ({n} launches)")
PY
