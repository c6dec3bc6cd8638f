#!/bin/bash
# tools/probes/ab_scan.sh <lib> ... — k_detect_scan alone (tools/probes/exp_scan.py) under alternate builds of libc8p_rx.so, three rounds on one box
for i in 1 2 3; do for l in "$@"; do echo "$l $(C8P_LIB=$l EXP_SCAN_REPS=20 python tools/probes/exp_scan.py 2>/dev/null)"; done; done
