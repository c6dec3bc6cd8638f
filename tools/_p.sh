export TMPDIR=/tmp
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
n=$(echo $c | tr ' ' '+')
EXP_SCAN_REPS=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$n -o pmc --output-format csv -- python $PWD/tools/exp_scan.py > /tmp/raw.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_$n/**/*counter_collection.csv",recursive=True)[0]
per=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("k_detect_scan"): per[int(r["Dispatch_Id"])][r["Counter_Name"]]+=float(r["Counter_Value"])
for d in sorted(per): print(d, {k: round(v) for k,v in per[d].items()})
PY
done
