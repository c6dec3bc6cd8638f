#!/bin/bash
# tools/sanitize_tsan.sh — ThreadSanitizer over the host library's threads (host/c8p_rawchain.cc: three stage threads, hand-over queues,
# the live session; host/c8p_rx_chain.cc and the scheduler harness; the blocks' latency timer) with tests/fake_c8p as the engine, driven
# from plain C (python under TSan is too slow): examples/rx_live.c on three capture files (complex64 and sc16), tools/tsan/blocks_driver.c
# (the rx_chain block, run and run_live), tools/tsan/chain_driver.c (demod -> decode on two threads + their latency timers, plain and fused).
# Rebuilds the CPU-side libraries with -fsanitize=thread, runs, restores the normal builds.  Prints the number of reports.
set -u
cd "$(dirname "$0")/.."
TS="-fsanitize=thread -fno-omit-frame-pointer -g"
W=/tmp/c8p_tsan
mkdir -p $W
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import numpy as np
from test_c_example import live_streams_of, sc16_files
import pathlib
xs, want = live_streams_of(900)
n = min(len(x) for x in xs)
for i, x in enumerate(xs):
    x.astype(np.complex64).tofile("/tmp/c8p_tsan/radio%d.c64" % i)
files, scale = sc16_files(pathlib.Path("/tmp/c8p_tsan"), xs)
open("/tmp/c8p_tsan/scale.txt", "w").write(repr(float(scale)))
print("captures: %d streams, %d PDUs expected" % (len(xs), sum(len(w) for w in want)))
# the sweep batch (every format / MCS of one antenna) at the signal -> demod boundary, for chain_driver.c
import struct
from c8p_cases import sweep_batch
b, _ = sweep_batch()
pk, iq, chan = b.arrays()
with open("/tmp/c8p_tsan/packets.bin", "wb") as f:
    k = 0
    for p in pk:
        if int(p["n_ant"]) != 1:
            continue
        n = int(p["n_items"])
        f.write(struct.pack("<Iiiiifff", n, int(p["lsig_mcs"]), int(p["lsig_len"]), int(p["lsig_nsamp"]), int(p["seq"]), float(p["cfo"]), float(p["snr"]), float(p["rssi"])))
        f.write(chan[int(p["chan_off"]):int(p["chan_off"]) + 64].astype(np.complex64).tobytes())
        f.write(iq[int(p["iq_off"][0]):int(p["iq_off"][0]) + n].astype(np.complex64).tobytes())
        k += 1
print("chain_driver: %d one-antenna packets" % k)
PY
make -s -C oracle clean && make -s -C oracle CFLAGS="-O1 -std=c11 -fPIC -ffp-contract=off -fno-fast-math $TS"
make -s -C gr-ieee80211_amd/host clean && make -s -C gr-ieee80211_amd/host CXXFLAGS="-O1 -std=c++17 -fPIC $TS"
( cd tests/fake_c8p && gcc -O1 -std=c11 -fPIC -ffp-contract=off $TS -I../../include -I../../oracle -shared -o libc8p_fake.so \
    fake_c8p.c ../../oracle/c8p_oracle.c ../../oracle/c8p_oracle_detect.c ../../oracle/c8p_oracle_tx.c -lm )
gcc -O1 -std=c11 -D_DEFAULT_SOURCE $TS -Iinclude -Igr-ieee80211_amd/host examples/rx_live.c -o $W/rx_live -Lgr-ieee80211_amd -lc8p_blocks -ldl -Wl,-rpath,$PWD/gr-ieee80211_amd
gcc -O1 -std=c11 $TS tools/tsan/blocks_driver.c -o $W/blocks_driver -Lgr-ieee80211_amd -lc8p_blocks -ldl -Wl,-rpath,$PWD/gr-ieee80211_amd
gcc -O1 -std=c11 $TS tools/tsan/chain_driver.c -o $W/chain_driver -Lgr-ieee80211_amd -lc8p_blocks -ldl -Wl,-rpath,$PWD/gr-ieee80211_amd
export TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1"
FAKE=$PWD/tests/fake_c8p/libc8p_fake.so
: > $W/err.txt
timeout 600 $W/rx_live -l $FAKE -r 7000 $W/radio0.c64 $W/radio1.c64 $W/radio2.c64 2>> $W/err.txt | tail -1
timeout 600 $W/rx_live -l $FAKE -r 5000 -s $(cat $W/scale.txt) $W/radio0.sc16 $W/radio1.sc16 $W/radio2.sc16 2>> $W/err.txt | tail -1
timeout 600 $W/blocks_driver $FAKE 6000 $W/radio0.c64 $W/radio1.c64 $W/radio2.c64 2>> $W/err.txt | tail -2
timeout 600 $W/chain_driver $FAKE $W/packets.bin 0 2>> $W/err.txt | tail -1
timeout 600 $W/chain_driver $FAKE $W/packets.bin 1 2>> $W/err.txt | tail -1
timeout 600 $W/chain_driver $FAKE $W/packets.bin -1 2>> $W/err.txt | tail -1     # nothing said: fused_auto decides (fused, lean tags)
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $W/err.txt)"
grep -A6 'WARNING: ThreadSanitizer' $W/err.txt | head -40
make -s -C oracle clean && make -s -C oracle
make -s -C gr-ieee80211_amd/host clean && make -s -C gr-ieee80211_amd/host
rm -f tests/fake_c8p/libc8p_fake.so && make -s -C tests/fake_c8p
