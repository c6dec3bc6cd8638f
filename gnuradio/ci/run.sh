#!/bin/sh
# gnuradio/ci/run.sh — the flowgraph contract test of SURVEY.md §8 f-3, on a machine with GNU Radio 3.10, the module of gnuradio/
# installed (gnuradio/ci/build.sh) and an MI355X:   one command, from the repository root.
set -eu
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cd "$HERE"
export C8P_RX_LIB="$HERE/gr-ieee80211_amd/libc8p_rx.so"
exec python3 -m pytest tests/test_gnuradio_flowgraph.py tests/test_gnuradio_packaging.py -m "gpu or not gpu" -q -rs
