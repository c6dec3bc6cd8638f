#!/bin/bash
# tools/ab.sh <libA> <libB> — alternate two builds of libc8p_rx.so on the same box (bench.py, no CPU / front-end legs)
for i in 1 2 3; do
  for l in "$1" "$2"; do
    C8P_LIB=$l python bench.py --no-cpu --no-frontend 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); x=j['roofline']['exclusive_avg_launch_ms']
print('$l', round(j['value']), round(j['ms_per_step'],4), 'excl', x['classify'], x['demod_sym'], x['viterbi_fwd'], x['finish'])"
  done
done
