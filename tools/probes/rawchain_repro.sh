#!/bin/bash
# tools/probes/rawchain_repro.sh <tag> — the raw-stream legs of bench.py on THIS lease under the switches VERDICT r3 asked about:
# default, ticket numbering forced (C8P_DET_TICKETS=1), and the runtime's hardware-queue count varied (GPU_MAX_HW_QUEUES).
# One JSON line per variant into gpurun_out/repro_<tag>.jsonl; tools/probes/rawchain_repro_table.py turns the files into the table.
tag=${1:-x}
out=gpurun_out/repro_$tag.jsonl
mkdir -p gpurun_out; : > $out
run() {   # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-configs --no-stream --no-host-legs --no-cpu 2> gpurun_out/repro_$tag.$name.err \
    | python -c "
import json, sys
j = json.loads(sys.stdin.readline())
c = j['frontend_detect']['raw_stream_to_pdus']
keep = {k: c.get(k) for k in ('packets_per_s', 'ms_per_segment', 'first_segment_ms', 'whole_run_packets_per_s', 'arena_growths_in_run', 'lookback_giveups', 'ticket_scans', 'stage_ms', 'host_thread_s', 'warmup', 'error')}
p = c.get('pcie_inclusive') or {}
l = c.get('live_streams') or {}
print(json.dumps({'variant': '$name', 'headline': j['value'], 'detect_ms_per_call': j['frontend_detect']['ms_per_call'], 'chain': keep,
                  'pcie': {k: p.get(k) for k in ('packets_per_s', 'ms_per_segment', 'GB_per_s_over_pcie', 'first_segment_ms', 'arena_growths_in_run', 'error')},
                  'live': {k: l.get(k) for k in ('streams', 'x_realtime_each', 'streams_at_realtime', 'pcie_GBps', 'ms_per_round', 'packets_per_s', 'lookback_giveups', 'error')}}))
" >> $out
}
run default C8P_X=0
run tickets C8P_DET_TICKETS=1
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run default_again C8P_X=0
cat $out | python -c "
import json, sys
for ln in sys.stdin:
    j = json.loads(ln)
    print('%-14s headline %.2f M  chain %.2f M/s (%.2f ms/seg, first %.1f ms, growths %s, giveups %s)  pcie %.2f M/s %.1f GB/s  live %.0f streams' % (
        j['variant'], j['headline'] / 1e6, (j['chain']['packets_per_s'] or 0) / 1e6, j['chain']['ms_per_segment'] or 0, j['chain']['first_segment_ms'] or 0,
        j['chain']['arena_growths_in_run'], j['chain']['lookback_giveups'], (j['pcie'].get('packets_per_s') or 0) / 1e6, j['pcie'].get('GB_per_s_over_pcie') or 0,
        j['live'].get('streams_at_realtime') or 0))
"
