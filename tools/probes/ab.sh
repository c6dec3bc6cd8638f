#!/bin/bash
# tools/probes/ab.sh <lib> [<lib> ...] — alternate builds of libc8p_rx.so on the same box (bench.py --no-check: kernel-variant timing
# experiments, the line is INVALID as a result), three rounds; prints packets/s, ms per step and the exclusive kernel times
for i in 1 2 3; do
  for l in "$@"; do
    C8P_LIB=$l python bench.py --no-cpu --no-frontend --no-stream --no-host-legs --no-sensitivity --detail /tmp/ab_detail.json --no-check --no-configs --steps 100 --repeats 3 --min-timed-s 0.6 > /dev/null 2>&1; python -c "
import json,sys; j=json.load(open('/tmp/ab_detail.json')); x=j['roofline']['exclusive_avg_launch_ms']
print('$l', round(j['value']), round(j['ms_per_step'],4), 'excl', x['classify'], x['demod_sym'], x['viterbi_fwd'], x['finish'])"
  done
done
