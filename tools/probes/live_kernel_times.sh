#!/bin/bash
# tools/probes/live_kernel_times.sh "<live_streams.py argument>" VAR=a VAR=b ... — on the GPU box: for every environment setting, one rocprofv3
# --stats run of tools/probes/live_streams.py and the average / minimum duration of the chain's main kernels in it (how much slower do they run in
# company than alone?)
R=$PWD
ARG=$1; shift
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  rm -rf /tmp/lkt
  echo "== $kv"
  env $kv timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/lkt -o ls --output-format csv -- python $R/tools/probes/live_streams.py $ARG 2>&1 | grep '"streams' | cut -c1-110
  F=$(find /tmp/lkt -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && python - "$F" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_rx", "k_classify", "k_finish", "k_copy_gather", "k_detect_scan")):
        print("   %-22s avg %8.1f us min %8.1f max %8.1f" % (n[:22], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
