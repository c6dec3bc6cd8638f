#!/bin/bash
# tools/probes/live_copy_cus.sh — the live ingest (64 streams of complex64, 128 of sc16, 128 of sc8) with the copy stream on 16 / 32 / 48 / 64 CUs
for cu in 16 32 48 64; do
  C8P_COPY_CUS=$cu python tools/probes/live_streams.py 64 128::sc16 128::sc8 2>/dev/null | python -c "
import json,sys
out=[]
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); out.append('%d streams: %.2f ms a round, %.0f at real time, %.1f GB/s' % (j['streams'], j['ms_per_round'], j['streams_at_realtime'], j['pcie_GBps']))
print('copy_cus $cu |', ' | '.join(out))"
done
