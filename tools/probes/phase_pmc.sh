#!/bin/bash
# tools/probes/phase_pmc.sh <out dir> — PMC passes (one counter set per run, --kernel-trace only) of k_rx_1ss_lds as it is (ab/libc8p_ph0.so),
# of its demodulator alone (ab/libc8p_ph1.so) and of its block loop alone (ab/libc8p_ph2.so): C8P_SLOTS=1, 4096 x cfg 2 per launch
O=${1:-gpurun_out/phase}; mkdir -p $O; export TMPDIR=/tmp
for k in 0 1 2; do
  C8P_LIB=$PWD/ab/libc8p_ph$k.so python tools/probes/phase_account.py > $O/account_ph$k.json 2> $O/account_ph$k.err
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
    n=$(echo $c | tr ' ' '+')
    C8P_LIB=$PWD/ab/libc8p_ph$k.so C8P_SLOTS=1 rocprofv3 --kernel-trace --pmc $c -d $O/ph${k}_$n -o pmc --output-format csv -- python $PWD/bench.py --steps 3 --warmup 1 --repeats 1 --min-timed-s 0 --no-cpu --no-check --legs none --detail /dev/null > $O/ph${k}_$n.log 2>&1
  done
done
python - $O <<'P'
import csv, glob, json, os, sys
from collections import defaultdict
O = sys.argv[1]
res = {}
for k in (0, 1, 2):
    cnt, dur = defaultdict(list), []
    for f in glob.glob(os.path.join(O, "ph%d_*" % k, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_rx_1ss_lds"):
                cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(O, "ph%d_GRBM_GUI_ACTIVE" % k, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_rx_1ss_lds"):
                dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res["ph%d" % k] = {c: sum(v) / len(v) for c, v in cnt.items()}
    res["ph%d" % k]["launches"] = max((len(v) for v in cnt.values()), default=0)
    res["ph%d" % k]["avg_us"] = sum(dur) / len(dur) / 1e3 if dur else None
    try:
        res["ph%d" % k]["account"] = json.loads(open(os.path.join(O, "account_ph%d.json" % k)).read().strip().splitlines()[-1])
    except Exception as e:
        res["ph%d" % k]["account"] = str(e)
json.dump(res, open(os.path.join(O, "phase_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
P
