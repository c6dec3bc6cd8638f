#!/bin/bash
# tools/probes/r6_lines.sh <tag> — on the GPU box: the driver's command (timed), the full record (--legs full), then the GPU tests;
# everything lands in gpurun_out/<tag>/ (copy bench*.json into profiles/<tag>/ afterwards)
T=${1:-r6a}; O=gpurun_out/$T; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_driver_cmd_detail.json > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time
python bench.py --legs full --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
wc -c $O/bench.json $O/bench_driver_cmd.json; cat $O/bench_driver_cmd.time
[ "${2:-}" = "notests" ] || { python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -5 $O/gpu_tests.txt; }
