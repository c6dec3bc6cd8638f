#!/bin/sh
# gnuradio/ci/build.sh <gr-ieee80211 checkout> — configure, build and install the complete module (needs GNU Radio 3.10, pybind11,
# UHD headers, ROCm's hipcc; no GPU).  This is the first time a compiler sees gnuradio/lib/*_impl.cc and c8p_gr_adapter.h.
set -eu
REF=${1:?usage: build.sh <path to a cloud9477/gr-ieee80211 checkout>}
HERE=$(cd "$(dirname "$0")/../.." && pwd)
make -C "$HERE/gr-ieee80211_amd/csrc"                       # libc8p_rx.so (HIP, gfx950)
make -C "$HERE/oracle"                                      # the tests' checker
cmake -S "$HERE/gnuradio" -B "$HERE/build/gnuradio" -DC8P_REFERENCE_TREE="$REF" -DCMAKE_INSTALL_PREFIX=/usr
cmake --build "$HERE/build/gnuradio" -j"$(nproc)"
cmake --install "$HERE/build/gnuradio"
ldconfig || true
python3 -c "from gnuradio import ieee80211; print('ieee80211 module:', [k for k in ('trigger','sync','signal','signal2','demod','demod2','decode') if hasattr(ieee80211, k)])"
