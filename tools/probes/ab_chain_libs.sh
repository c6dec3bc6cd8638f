#!/bin/bash
# tools/probes/ab_chain_libs.sh <lib> [<lib> ...] — the demod -> decode block chain (tools/probes/chain_custom.py: custom buffers, max_batch 512,
# plain and fused) under alternate builds of libc8p_rx.so on the same box, three rounds
for i in 1 2 3; do
  for l in "$@"; do
    C8P_LIB=$PWD/$l python tools/probes/chain_custom.py 512 custom both 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('$l', 'plain', j.get('plain'), j.get('plain_best'), 'fused', j.get('fused'), j.get('fused_best'))"
  done
done
