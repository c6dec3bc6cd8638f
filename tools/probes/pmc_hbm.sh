#!/bin/bash
# tools/probes/pmc_hbm.sh <tag> — run ON the GPU box: HBM traffic per kernel of one c8p_rx step (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass
# each, kernels serialised: C8P_SLOTS=1), raw csv under gpurun_out/pmc_<tag>/; prints bytes per launch (FETCH doubled per MI355X_MICROARCH.md,
# raw beside it).
set -u
TAG=${1:-x}; OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o pmc --output-format csv -- python $PWD/bench.py --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --no-frontend --no-stream --no-host-legs --no-configs > $OUT/$c.log 2>&1
done
python - $OUT <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    tot[c] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
print("%-28s %8s %14s %14s %14s" % ("kernel", "launches", "FETCH raw MB", "FETCH x2 MB", "WRITE MB"))
sf = sw = sf2 = 0
for k in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"])):
    if not k.startswith("k_"): continue
    f, n = tot["FETCH_SIZE"].get(k, (0, 0)); w, _ = tot["WRITE_SIZE"].get(k, (0, 0))
    print("%-28s %8d %14.1f %14.1f %14.1f" % (k, n, f * 1024 / 1e6, 2 * f * 1024 / 1e6, w * 1024 / 1e6))   # counters are in KB
    sf += f * 1024 / 1e6; sw += w * 1024 / 1e6
print("%-28s %8s %14.1f %14.1f %14.1f" % ("sum", "", sf, 2 * sf, sw))
P
