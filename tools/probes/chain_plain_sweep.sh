#!/bin/bash
# tools/probes/chain_plain_sweep.sh — the PLAIN demod -> decode chain (every LLR item written: 35 KB down + 57 KB up per packet) under more engine
# slots (batches in flight) and deeper gather queues: is the link's idle 40 % a pipeline that is too shallow?
for s in 3 4 6; do for d in 2 4; do for mb in 512 256; do
  C8P_SLOTS=$s C8P_GATHER_DEPTH=$d python tools/probes/chain_custom.py $mb custom plain 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('slots $s depth $d max_batch $mb plain', j.get('plain'), j.get('plain_best'), j.get('plain_demod'))"
done; done; done
