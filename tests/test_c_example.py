"""examples/rx_file.c: the C ABI used from plain C — capture file -> c8p_detect -> c8p_signal -> c8p_rx -> PDUs."""
import os
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "rx_file")


def build():
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "rx_file.c"), "-o", EXE, "-L" + os.path.join(ROOT, "gr-ieee80211_amd"),
                           "-lc8p_rx", "-Wl,-rpath," + os.path.join(ROOT, "gr-ieee80211_amd")])


def test_example_compiles_as_plain_c11_against_the_header():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_example_turns_a_capture_file_into_pdus(tmp_path):
    from c8p_testlib import abi
    from gr_ieee80211_amd import engine
    from test_tx import psdu_of, tx_batch
    build()
    rng = np.random.default_rng(12)
    specs = []
    for fmt, mcs in ((0, 0), (0, 7), (1, 3), (1, 7), (2, 0), (2, 5), (2, 8), (0, 4), (2, 7)):
        body = bytes(rng.integers(0, 256, size=int(rng.integers(40, 900)), dtype=np.uint8))
        mp = body + zlib.crc32(body).to_bytes(4, "little")
        specs.append((fmt, mcs, mp, psdu_of(fmt, mp)))
    pk, arena = tx_batch(specs, gap=700, mult=0.3)
    pk["flags"] = abi.TX_NO_WINDOW
    e = engine.Engine(device=0)
    e.tx(pk, arena, tail_gap=800)
    e.sync()
    iq = e.fetch(abi.BUF_TX_IQ, np.complex64)
    e.close()
    iq = (iq + 6.0e-5 * (rng.normal(size=len(iq)) + 1j * rng.normal(size=len(iq)))).astype(np.complex64)
    cap = tmp_path / "capture.c64"
    iq.tofile(cap)
    r = subprocess.run([EXE, str(cap)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "%d packets, %d PDUs" % (len(specs), len(specs)), r.stdout
    pdus = [l for l in lines if l.startswith("pdu ")]
    for l, (fmt, mcs, mp, ps) in zip(pdus, specs):
        assert "%s mcs %d, %d bytes, crc32 correct" % (("legacy", "ht", "vht")[fmt], mcs, len(mp)) in l, l
        assert l.endswith("first bytes %02x %02x %02x %02x" % tuple(mp[:4])), l


# ------------------------------------------------------------------------------------------------ examples/rx_live.c
LIVE = os.path.join(ROOT, "examples", "rx_live")


def build_live():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gr-ieee80211_amd", "host")])
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-D_DEFAULT_SOURCE", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "gr-ieee80211_amd", "host"), os.path.join(ROOT, "examples", "rx_live.c"), "-o", LIVE,
                           "-L" + os.path.join(ROOT, "gr-ieee80211_amd"), "-lc8p_blocks", "-ldl", "-Wl,-rpath," + os.path.join(ROOT, "gr-ieee80211_amd")])


def live_streams_of(seed):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_detect_frontend import CFOS, NAMES, SWEEP, stream
    order = [NAMES, NAMES[::-1], NAMES[2:6]]
    xs = [np.concatenate([stream(nm, CFOS[:len(nm)], snr=33.0, seed=seed + i)[0], np.zeros(900, np.complex64)]) for i, nm in enumerate(order)]
    return xs, [[SWEEP[k]["pdu"].tobytes() for k in nm] for nm in order]


def check_live_output(out, want, rounds_min):
    lines = out.strip().splitlines()
    got = {s: [] for s in range(len(want))}
    last = {}
    for ln in lines[:-1]:
        f = ln.split()
        s, item = int(f[3]), int(f[5].rstrip(":"))
        assert last.get(s, -1) < item, ln                    # per stream in stream order
        last[s] = item
        got[s].append((int(f[7]), ln.split("first bytes ")[1]))
    for s, ps in enumerate(want):
        assert [g[0] for g in got[s]] == [len(p) - 4 for p in ps], s
        assert [g[1] for g in got[s]] == ["%02x %02x %02x %02x" % tuple(p[3:7]) for p in ps], s
    tot = sum(len(p) for p in want)
    assert lines[-1].startswith("%d streams, " % len(want)) and lines[-1].endswith("%d packets, %d PDUs" % (tot, tot)), lines[-1]
    assert int(lines[-1].split()[2]) >= rounds_min


def test_live_example_compiles_as_plain_c11_and_runs_on_the_test_engine(tmp_path):
    """plain C against include/c8p_rx.h + host/c8p_rawchain.h; here with tests/fake_c8p (the C ABI on the oracle) as the engine
    library: three capture files = three radios, pushed out of step, every stream's PDUs in order"""
    build_live()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])
    xs, want = live_streams_of(900)
    files = []
    for i, x in enumerate(xs):
        f = tmp_path / ("radio%d.c64" % i)
        x.astype(np.complex64).tofile(f)
        files.append(str(f))
    r = subprocess.run([LIVE, "-l", os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so"), "-r", "7000"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    check_live_output(r.stdout, want, 3)


def sc16_files(tmp_path, xs):
    """the captures as the radios would send them (int16 I/Q pairs) and the scale that turns them back into floats"""
    scale = np.float32(1.02 * max(float(np.abs(x.view(np.float32)).max()) for x in xs) / 32767.0)
    files = []
    for i, x in enumerate(xs):
        f = tmp_path / ("radio%d.sc16" % i)
        np.clip(np.rint(x.astype(np.complex64).view(np.float32) / scale), -32768, 32767).astype(np.int16).tofile(f)
        files.append(str(f))
    return files, scale


def test_live_example_takes_sc16_captures_on_the_test_engine(tmp_path):
    """-s scale: the files hold int16 I/Q pairs, pushed with C8P_MEM_PINNED | C8P_MEM_SC16 out of c8p_host_alloc memory"""
    build_live()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "fake_c8p")])
    xs, want = live_streams_of(900)
    files, scale = sc16_files(tmp_path, xs)
    r = subprocess.run([LIVE, "-l", os.path.join(ROOT, "tests", "fake_c8p", "libc8p_fake.so"), "-r", "7000", "-s", repr(float(scale))] + files,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    check_live_output(r.stdout, want, 3)


@pytest.mark.gpu
def test_live_example_takes_sc16_captures_on_the_engine(tmp_path):
    build_live()
    xs, want = live_streams_of(910)
    files, scale = sc16_files(tmp_path, xs)
    r = subprocess.run([LIVE, "-l", os.path.join(ROOT, "gr-ieee80211_amd", "libc8p_rx.so"), "-r", "9000", "-s", repr(float(scale))] + files,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    check_live_output(r.stdout, want, 3)


@pytest.mark.gpu
def test_live_example_on_the_engine(tmp_path):
    build_live()
    xs, want = live_streams_of(910)
    files = []
    for i, x in enumerate(xs):
        f = tmp_path / ("radio%d.c64" % i)
        x.astype(np.complex64).tofile(f)
        files.append(str(f))
    r = subprocess.run([LIVE, "-l", os.path.join(ROOT, "gr-ieee80211_amd", "libc8p_rx.so"), "-r", "9000"] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    check_live_output(r.stdout, want, 3)
