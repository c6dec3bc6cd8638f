for i in 1 2; do for mb in 256 512 1024; do
  python tools/probes/chain_custom.py $mb custom fused 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('max_batch $mb fused', j.get('fused'), j.get('fused_best'), j.get('fused_demod'))"
done; done
for q in 4 8; do GPU_MAX_HW_QUEUES=$q python tools/probes/chain_custom.py 512 custom fused 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    if ln.startswith('{'):
        j=json.loads(ln); print('hwq $q fused', j.get('fused'), j.get('fused_best'))"; done
