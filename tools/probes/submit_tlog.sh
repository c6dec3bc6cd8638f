#!/bin/bash
# tools/probes/submit_tlog.sh — where the host time of a 4096-packet c8p_rx submit goes (C8P_SUBMIT_TLOG=1, printed by the engine at exit)
C8P_SUBMIT_TLOG=1 python bench.py --steps 200 --repeats 3 --min-timed-s 0 --no-cpu --legs none --detail /tmp/tlog_detail.json 2>&1 >/dev/null | grep "c8p submit"
