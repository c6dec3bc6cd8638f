#!/bin/bash
for l in 0 12000 25000 52000; do
  echo -n "lds $l: "
  C8P_PRE_LDS=$l python tools/run_detect.py 2>/dev/null | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read().strip()); print(d['kernel_ms'], round(d['presiso_roofline']['frac'],3))"
done
