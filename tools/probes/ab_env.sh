#!/bin/bash
# tools/probes/ab_env.sh VAR=a VAR=b ... — alternate settings of ONE environment switch of the engine on the same box (bench.py, cfg 2), three rounds;
# prints packets/s, ms per step and the exclusive kernel times (classify, demod, forward pass, finish)
for i in 1 2 3; do
  for kv in "$@"; do
    env $kv python bench.py --no-cpu --no-frontend --no-stream --no-host-legs --no-sensitivity --detail /tmp/ab_detail.json --no-configs --steps 100 --repeats 3 --min-timed-s 0.6 > /dev/null 2>&1; python -c "
import json,sys; j=json.load(open('/tmp/ab_detail.json')); x=j['roofline']['exclusive_avg_launch_ms']
print('$kv', round(j['value']), round(j['ms_per_step'],4), 'excl', x['classify'], x['demod_sym'], x['viterbi_fwd'], x['finish'])"
  done
done
