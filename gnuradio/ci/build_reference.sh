#!/bin/sh
# gnuradio/ci/build_reference.sh <gr-ieee80211 checkout> [prefix] — build and install the UNMODIFIED reference module under a prefix of
# its own (default /opt/c8p-ref), beside the module of gnuradio/ in the system prefix.  Both define gr::ieee80211::demod etc. and the
# Python module gnuradio.ieee80211, so they can never share a process: the side-by-side test runs gnuradio/ci/dump_chain.py once per
# installation (the reference through --module-dir + LD_LIBRARY_PATH) and compares the dumps.  Nothing of the reference is changed
# or copied: its own CMake project is configured where it lies.
set -eu
REF=${1:?usage: build_reference.sh <path to a cloud9477/gr-ieee80211 checkout> [install prefix]}
PREFIX=${2:-/opt/c8p-ref}
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cmake -S "$REF" -B "$HERE/build/reference" -DCMAKE_INSTALL_PREFIX="$PREFIX" -DCMAKE_BUILD_TYPE=Release
cmake --build "$HERE/build/reference" -j"$(nproc)"
cmake --install "$HERE/build/reference"
MOD=$(find "$PREFIX" -type d -path "*gnuradio/ieee80211" | head -1)
echo "reference module: $MOD"
echo "C8P_REF_PREFIX=$PREFIX" > "$HERE/gnuradio/ci/reference.env"
echo "C8P_REF_MODULE_DIR=$(dirname "$MOD")" >> "$HERE/gnuradio/ci/reference.env"
