#!/usr/bin/env python3
"""tools/probes/count_demod_isa.py — static instruction counts of the pieces of one OFDM symbol's demodulation (tools/probes/demod_probe.hip: one
small kernel per piece, compiled with the engine's flags; hipcc cross-compiles without a GPU).  DESIGN.md §9 quotes the table:
it is the account of what k_rx_1ss_lds spends outside the Viterbi forward pass (measured: 69 M of its 313 M VALU wave-instructions
per 4096 packets = 368 per symbol)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt "
         "-mllvm -disable-promote-alloca-to-lds=1").split()


def main():
    out = os.path.join(tempfile.mkdtemp(), "demod_probe.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gr-ieee80211_amd", "csrc"),
                                                             "-S", "--cuda-device-only", "-o", out, os.path.join(ROOT, "tools", "demod_probe.hip")],
                          stderr=subprocess.DEVNULL)
    s = open(out).read()
    print("%-11s %6s %6s %6s %5s %5s %5s" % ("piece", "VALU", "f64", "DS", "SALU", "VMEM", "SMEM"))
    for name in ("p_empty", "p_fft", "p_cdiv", "p_cabs", "p_cdivf", "p_pilots", "p_scatter6", "p_sym"):
        body = re.search(r"^%s:.*?\.end_amdhsa_kernel" % name, s, re.S | re.M).group(0)
        ins = [ln.split()[0] for ln in body.splitlines() if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        valu = [i for i in ins if i.startswith("v_")]
        print("%-11s %6d %6d %6d %5d %5d %5d" % (name, len(valu), sum("f64" in i for i in valu), sum(i.startswith("ds_") for i in ins),
                                                 sum(i.startswith("s_") and not i.startswith(("s_load", "s_waitcnt", "s_nop", "s_endpgm")) for i in ins),
                                                 sum(i.startswith(("global_", "flat_", "buffer_")) for i in ins), sum(i.startswith("s_load") for i in ins)))
    print("(p_sym = the whole symbol with all five constellations' branches present: static, not per-symbol; p_empty = the harness alone)")


if __name__ == "__main__":
    sys.exit(main())
