#!/bin/bash
# tools/probes/chain_copy_cus.sh [cus ...] — the demod -> decode block chain (plain and fused) with the engine's copy stream confined to
# that many CUs (C8P_COPY_CUS), alternated, three rounds
for i in 1 2 3; do for cu in "${@:-16 32}"; do
  C8P_COPY_CUS=$cu python tools/probes/chain_custom.py 512 custom both 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('copy_cus $cu plain', j.get('plain'), j.get('plain_best'), 'fused', j.get('fused'), j.get('fused_best'))"
done; done
