#!/bin/bash
# tools/probes/pmc_scan.sh <tag> — run ON the GPU box: k_detect_scan's instruction counts per wave (SQ_INSTS_VALU / SALU / LDS, SQ_WAVES; one
# rocprofv3 --pmc pass per counter pair) over the two streams of tools/probes/exp_scan.py: launches 1.. of the packet-dense bench stream first,
# then the noise-only one (every tile on the quiet path).
set -u
TAG=${1:-x}; OUT=gpurun_out/pmcscan_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
python tools/probes/exp_scan.py > $OUT/plain.json 2> $OUT/plain.err; cat $OUT/plain.json
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=$(echo $c | tr ' ' '_')
  EXP_SCAN_REPS=2 rocprofv3 --kernel-trace --pmc $c -d $OUT/$d -o pmc --output-format csv -- python $PWD/tools/probes/exp_scan.py > $OUT/$d.log 2>&1
done
python - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_detect_scan"):
            acc[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
for c, d in sorted(acc.items()):
    for k, v in d.items():
        h = len(v) // 2          # first half of the launches: packets, second half: noise
        print("%-18s %-16s packets %14.0f   noise %14.0f   (launches %d)" % (c, k, sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1), len(v)))
P
