#!/bin/bash
# tools/probes/r6_verify.sh <tag> [soak batches] [split batches] — on the GPU box, on the code as committed: the GPU tests, smoke(), then the randomized
# differential soak (tools/soak_random.py) with a seed no earlier round used; everything lands in gpurun_out/<tag>/
T=${1:-r6v}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python tools/soak_random.py ${2:-300} 2048 6061 > $O/soak_random.txt 2>&1; tail -2 $O/soak_random.txt
python tools/soak_random.py ${3:-40} 2048 6062 split > $O/soak_random_split.txt 2>&1; tail -2 $O/soak_random_split.txt
