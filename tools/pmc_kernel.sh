#!/bin/bash
# SQ counters of ONE kernel family under a kbench command (one --pmc pass per counter group; no trace domains beside them):
#   bash tools/pmc_kernel.sh <tag> <kernel-name substring> <kbench args...>
#   e.g. bash tools/pmc_kernel.sh pp_conv igemm_pp_kernel conv --cin 320 --cout 320 --hw 64 --images 16 --iters 3
set -u
TAG=$1; FILTER=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python tools/kbench.py "$@" > $OUT/run$i.log 2> $OUT/err$i.log || { echo "pass $i failed"; tail -3 $OUT/err$i.log; }
done
FILTER="$FILTER" OUTDIR="$OUT" python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: [0.0, 0])
flt, out = os.environ["FILTER"], os.environ["OUTDIR"]
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r['Kernel_Name']: continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
with open(out + "/summary.txt", "w") as fh:
    for k, (v, n) in sorted(acc.items()):
        line = f"{k:32s} {v / n:16.0f} per launch ({n} launches)"
        print(line); fh.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
