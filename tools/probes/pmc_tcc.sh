export TMPDIR=/tmp; OUT=gpurun_out/pmc_tcc_$1; mkdir -p $OUT
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_READ_sum TCC_WRITE_sum TCC_EA_RDREQ_sum" "TCC_EA_RDREQ_32B_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '+')
  C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $OUT/$n -o pmc --output-format csv -- python $PWD/bench.py --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --no-frontend --no-stream --no-host-legs --no-configs > $OUT/$n.log 2>&1
done
python - $OUT <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in sorted(acc[k].items())}, "(millions per launch)")
P
