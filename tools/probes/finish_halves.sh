#!/bin/bash
# tools/probes/finish_halves.sh <out dir> — VERDICT r5 item 7: does k_finish read the survivors out of the 256 MB Infinity Cache when it
# follows k_rx_1ss_lds on a HALF batch (2048 packets: 209 MB of survivors) instead of a whole one (4096: 418 MB)?  One slot (C8P_SLOTS=1:
# classify, rx, finish of a batch back to back on one stream), batches of 1024 / 2048 / 4096 / 8192 packets: k_finish's exclusive launch
# duration per packet (bench.py's HIP events), and its EA read requests / fetched bytes per packet (PMC passes, --kernel-trace only).
O=${1:-gpurun_out/finish_halves}; mkdir -p $O; export TMPDIR=/tmp
for b in 1024 2048 4096 8192; do
  C8P_SLOTS=1 python bench.py --batch $b --steps 60 --repeats 3 --min-timed-s 0 --no-cpu --legs none --detail $O/b$b.json > /dev/null 2> $O/b$b.err
done
for b in 2048 4096; do
  for c in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '+')
    C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_b${b}_$n -o pmc --output-format csv -- python $PWD/bench.py --batch $b --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --legs none --detail /dev/null > $O/pmc_b${b}_$n.log 2>&1
  done
done
python - $O <<'P'
import csv, glob, json, os, sys
from collections import defaultdict
O = sys.argv[1]
res = {"exclusive_launch": {}, "pmc_k_finish": {}}
for b in (1024, 2048, 4096, 8192):
    try:
        j = json.load(open(os.path.join(O, "b%d.json" % b)))
        x = j["roofline"]["exclusive_avg_launch_ms"]
        res["exclusive_launch"][b] = {"finish_ms": x["finish"], "finish_ns_per_packet": 1e6 * x["finish"] / b, "rx_ms": x["viterbi_fwd"], "rx_ns_per_packet": 1e6 * x["viterbi_fwd"] / b,
                                      "step_ms": j["ms_per_step"], "packets_per_s": j["value"]}
    except Exception as e:
        res["exclusive_launch"][b] = str(e)
for b in (2048, 4096):
    cnt, dur = defaultdict(list), []
    for f in glob.glob(os.path.join(O, "pmc_b%d_*" % b, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_finish"):
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res["pmc_k_finish"][b] = {c: sum(v) / len(v) / b for c, v in cnt.items()}      # per packet
json.dump(res, open(os.path.join(O, "finish_halves.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
P
