"""Register / LDS budgets of the hot kernels, from the compiler's own resource remarks (hipcc cross-compiles without a GPU).

The pipelined step keeps the launches of four batches resident at once, and its throughput follows how many decoder waves fit
next to the side kernels: measured on MI355X, a decoder build with 66 instead of 64 VGPRs (7 instead of 8 waves per SIMD) is
4-8 % slower per step although the kernel alone is faster, and 18 KB of LDS per decoder workgroup (the compiler's alloca
promotion) cost up to 6.5 %.  These budgets are therefore part of the design and guarded here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr-ieee80211_amd", "csrc")


@pytest.fixture(scope="module")
def usage():
    r = subprocess.run(["make", "-s", "-C", CSRC, "asm"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res, cur = {}, None
    for line in (r.stdout + r.stderr).splitlines():
        m = re.search(r"Function Name: _Z(\d+)(\w+)", line)
        if m:
            cur = res.setdefault(m.group(2)[:int(m.group(1))], {})       # Itanium mangling: <length><name>
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


def test_decoder_and_symbol_kernels_keep_eight_waves_per_simd(usage):
    lean, demod = usage["k_viterbi_lean"], usage["k_demod_sym"]
    assert lean["VGPRs"] <= 64 and lean["AGPRs"] == 0 and lean["Occupancy"] == 8, lean
    assert lean["LDS Size"] == 0, lean                      # no alloca promoted into LDS (-disable-promote-alloca-to-lds)
    assert lean["VGPRs Spill"] == 0 and lean["SGPRs Spill"] == 0, lean
    assert demod["VGPRs"] <= 64 and demod["Occupancy"] == 8 and demod["LDS Size"] == 0, demod
    assert demod["ScratchSize"] == 0 and demod["VGPRs Spill"] == 0, demod


def test_fused_rx_kernel_keeps_six_waves_per_simd(usage):
    """k_rx_1ss (symbol demodulation + forward pass in one wave): decoder state parked in LDS during the demodulation, lane constants packed"""
    rx = usage["k_rx_1ss"]
    assert rx["VGPRs"] <= 80 and rx["AGPRs"] == 0 and rx["Occupancy"] >= 6, rx
    assert rx["LDS Size"] <= 8 * 1024, rx
    assert rx["ScratchSize"] <= 96, rx                     # loop-invariant values around the loop, nothing inside the rate-5/6 path


def test_on_chip_llr_kernel_has_no_scratch_at_six_waves(usage):
    """k_rx_1ss_lds (c8p_rx's default: LLRs from the demodulator to the decoder through LDS): 80 VGPRs = 6 waves per SIMD with NOTHING
    spilled — a dword spilled around the decoder is 64 x 4 bytes per OFDM symbol and packet of scratch traffic, as much as the LLR
    stream this kernel keeps out of HBM — and an LDS footprint that lets six workgroups share a CU (160 KB)"""
    rx = usage["k_rx_1ss_lds"]
    assert rx["VGPRs"] <= 80 and rx["AGPRs"] == 0 and rx["Occupancy"] >= 6, rx
    assert rx["ScratchSize"] == 0 and rx["VGPRs Spill"] == 0 and rx["SGPRs Spill"] == 0, rx
    assert rx["LDS Size"] <= 22 * 1024, rx                 # x 6 workgroups per CU = 128 KB of 160 KB


def test_side_kernels_stay_inside_their_budgets(usage):
    assert usage["k_classify"]["VGPRs"] <= 72, usage["k_classify"]
    assert usage["k_finish"]["LDS Size"] <= 24 * 1024 and usage["k_finish"]["ScratchSize"] == 0, usage["k_finish"]
    assert usage["k_tx_sym"]["ScratchSize"] <= 16
    # the one-pass detection scan: nothing spilled, and an LDS footprint that lets five workgroups share a CU
    ds = usage["k_detect_scan"]
    assert ds["ScratchSize"] == 0 and ds["VGPRs"] <= 96 and ds["LDS Size"] <= 30 * 1024, ds


def test_generated_block_loop_is_current(tmp_path):
    """csrc/c8p_lean_asm.h is generated (tools/gen_lean_asm.py) and committed: the committed file must be what the generator writes"""
    import sys
    out = tmp_path / "c8p_lean_asm.h"
    env = dict(os.environ, C8P_GEN_OUT=str(out))
    env.pop("C8P_GEN_EXP", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_lean_asm.py")], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert out.read_text() == open(os.path.join(CSRC, "c8p_lean_asm.h")).read()
    out4 = tmp_path / "c8p_lean4_asm.h"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_lean4_asm.py")], capture_output=True, text=True,
                       env=dict(os.environ, C8P_GEN_OUT4=str(out4)), timeout=120)
    assert r.returncode == 0, r.stderr
    assert out4.read_text() == open(os.path.join(CSRC, "c8p_lean4_asm.h")).read()
    out2 = tmp_path / "c8p_lean2_asm.h"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_lean2_asm.py")], capture_output=True, text=True,
                       env=dict(os.environ, C8P_GEN_OUT2=str(out2)), timeout=120)
    assert r.returncode == 0, r.stderr
    assert out2.read_text() == open(os.path.join(CSRC, "c8p_lean2_asm.h")).read()
