#!/bin/bash
# tools/probes/pmc_scan2.sh <tag> — k_detect_scan's wave-cycle account (busy / waiting / by instruction class), same two streams as pmc_scan.sh
set -u
TAG=${1:-x}; OUT=gpurun_out/pmcscan2_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_WAVE32"; do
  d=$(echo $c | tr ' ' '_')
  EXP_SCAN_REPS=2 rocprofv3 --kernel-trace --pmc $c -d $OUT/$d -o pmc --output-format csv -- python $PWD/tools/probes/exp_scan.py > $OUT/$d.log 2>&1 || echo "pass $d failed"
done
python - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].split("(")[0].startswith("k_detect_scan"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    h = len(v) // 2
    print("%-24s packets %16.0f   noise %16.0f" % (c, sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1)))
P
