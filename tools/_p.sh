export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o raw --output-format csv -- python $PWD/tools/exp_rawchain.py > /tmp/raw.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/prof/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith("k_"): print(r["Name"][:30].ljust(30), r["Calls"].rjust(5), "%10.1f us avg" % (float(r["AverageNs"])/1e3))
PY
