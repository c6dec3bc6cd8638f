#!/bin/bash
# tools/sanitize_cpu.sh — rebuild the CPU-side libraries (oracle, fake backend, C++ block mirror + harness) with
# AddressSanitizer + UndefinedBehaviorSanitizer and run the CPU test-suite under them, then restore the normal builds.
# (GPU sanitizers are not available on the pool; the host-side state machines and the oracle are what this covers.)
set -u
cd "$(dirname "$0")/.."
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
LIBDIR=$(dirname "$(gcc -print-file-name=libasan.so)")
make -s -C oracle clean && make -s -C oracle CFLAGS="-O1 -std=c11 -fPIC -ffp-contract=off -fno-fast-math $SAN"
make -s -C gr-ieee80211_amd/host clean && make -s -C gr-ieee80211_amd/host CXXFLAGS="-O1 -std=c++17 -fPIC $SAN"
( cd tests/fake_c8p && gcc -O1 -std=c11 -fPIC -ffp-contract=off $SAN -I../../include -I../../oracle -shared -o libc8p_fake.so \
    fake_c8p.c ../../oracle/c8p_oracle.c ../../oracle/c8p_oracle_detect.c ../../oracle/c8p_oracle_tx.c -lm )
LD_PRELOAD=$LIBDIR/libasan.so:$LIBDIR/libubsan.so ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_kernel_resources.py --deselect tests/test_shard_gloo.py > /tmp/c8p_san.log 2>&1
echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' /tmp/c8p_san.log)"; tail -1 /tmp/c8p_san.log
make -s -C oracle clean && make -s -C oracle
make -s -C gr-ieee80211_amd/host clean && make -s -C gr-ieee80211_amd/host
rm -f tests/fake_c8p/libc8p_fake.so && make -s -C tests/fake_c8p
