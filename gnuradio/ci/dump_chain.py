#!/usr/bin/env python3
"""gnuradio/ci/dump_chain.py — run rx.grc's receive chain on an IQ file inside GNU Radio and dump what crosses the demod -> decode edge.

    dump_chain.py run --iq FILE.c64 [--iq2 FILE2.c64] --out DIR [--module-dir DIR]
    dump_chain.py compare DIR_REFERENCE DIR_OURS [--json OUT.json]

`run` builds file source -> presiso -> trigger -> sync -> signal[2] -> demod[2] -> decode (examples/rx.grc:753-767, rx2.grc:676-692) with
WHICHEVER `gnuradio.ieee80211` the process finds (--module-dir prepends a directory to gnuradio.__path__: the unmodified reference
module installed under its own prefix — gnuradio/ci/build_reference.sh — next to the module of gnuradio/ in the system prefix; the two
define the same C++ classes and cannot share a process), taps the LLR edge with a Python sink (items + tags: a second reader, so a demod
of this repository runs in plain mode and writes every item) and stores
    DIR/llr.f32      the float32 item stream of the edge
    DIR/tags.json    [[absolute offset, key, value], ...] in stream order (value: int, float, or [[re, im], ...] for c32vectors)
    DIR/pdus.json    the blobs decode published, hex
    DIR/stdout.txt   what the blocks printed (decode's debug lines)
`compare` checks ours against the reference's: PDUs identical, tags identical (integers and pass-through floats exactly, computed floats
to 1e-4 relative; keys beginning with c8p_ are this module's own and reported, not compared), LLR items |difference| <= 1e-5 — the bound
BASELINE.json states — and prints / stores the largest differences.  This is the comparison that pins the LLR stream to the reference's
C++ (lib/demod_impl.cc:279-314 through FFTW3f); tests/test_gnuradio_flowgraph.py::test_reference_cxx_side_by_side drives it.

The comparison half needs only numpy and is exercised by the CPU suite (tests/test_gnuradio_packaging.py) on dumps written from the oracle."""
import argparse
import json
import os
import sys

import numpy as np

LLR_TOL = 1e-5
COMPUTED_FLOAT_KEYS = ("sssnr0", "sssnr1")          # computed by demod (double accumulate + log10, demod_impl.cc:489-504): compared to 1e-4 relative


def py_value(pmt, v):
    if pmt.is_integer(v):
        return int(pmt.to_long(v))
    if pmt.is_real(v):
        return float(pmt.to_double(v))
    if pmt.is_c32vector(v):
        return [[float(z.real), float(z.imag)] for z in pmt.c32vector_elements(v)]
    return str(v)


def run(args):
    import gnuradio
    if args.module_dir:
        gnuradio.__path__.insert(0, args.module_dir)
    from gnuradio import blocks, gr
    import pmt
    from gnuradio import ieee80211

    class edge_tap(gr.sync_block):
        """a second reader of the LLR edge: every item and every tag, in order"""

        def __init__(self):
            gr.sync_block.__init__(self, "c8p_edge_tap", in_sig=[np.float32], out_sig=None)
            self.items, self.tags = [], []

        def work(self, input_items, output_items):
            x = input_items[0]
            self.items.append(x.copy())
            for t in self.get_tags_in_window(0, 0, len(x)):
                self.tags.append([int(t.offset), pmt.symbol_to_string(t.key), py_value(pmt, t.value)])
            return len(x)

    class presiso(gr.hier_block2):
        """examples/presiso.grc rebuilt from the stock blocks it names"""

        def __init__(self):
            gr.hier_block2.__init__(self, "presiso", gr.io_signature(1, 1, gr.sizeof_gr_complex),
                                    gr.io_signature.makev(2, 2, [gr.sizeof_float, gr.sizeof_gr_complex]))
            delay, mulc = blocks.delay(gr.sizeof_gr_complex, 16), blocks.multiply_conjugate_cc(1)
            avg_c, avg_f = blocks.moving_average_cc(48, 1, 4000, 1), blocks.moving_average_ff(64, 1, 4000, 1)
            mag, mag2, div = blocks.complex_to_mag(1), blocks.complex_to_mag_squared(1), blocks.divide_ff(1)
            self.connect((self, 0), (mag2, 0), (avg_f, 0), (div, 1))
            self.connect((self, 0), (delay, 0), (mulc, 0))
            self.connect((self, 0), (mulc, 1))
            self.connect((mulc, 0), (avg_c, 0), (mag, 0), (div, 0))
            self.connect((div, 0), (self, 0))
            self.connect((avg_c, 0), (self, 1))

    os.makedirs(args.out, exist_ok=True)
    tb = gr.top_block("c8p side by side", catch_exceptions=False)
    files = [args.iq] + ([args.iq2] if args.iq2 else [])
    srcs = [blocks.file_source(gr.sizeof_gr_complex, f, False) for f in files]
    pre, trig, sync = presiso(), ieee80211.trigger(), ieee80211.sync()
    dec, dbg, tap = ieee80211.decode(True), blocks.message_debug(), edge_tap()
    tb.connect((srcs[0], 0), (pre, 0))
    tb.connect((pre, 0), (trig, 0))
    tb.connect((trig, 0), (sync, 0))
    tb.connect((pre, 1), (sync, 1))
    tb.connect((srcs[0], 0), (sync, 2))
    if args.iq2:
        sig, dem = ieee80211.signal2(), ieee80211.demod2()
        tb.connect((sync, 0), (sig, 0))
        tb.connect((srcs[0], 0), (sig, 1))
        tb.connect((srcs[1], 0), (sig, 2))
        tb.connect((sig, 0), (dem, 0))
        tb.connect((sig, 1), (dem, 1))
    else:
        sig, dem = ieee80211.signal(), ieee80211.demod(0, 2)
        tb.connect((sync, 0), (sig, 0))
        tb.connect((srcs[0], 0), (sig, 1))
        tb.connect((sig, 0), (dem, 0))
    tb.connect((dem, 0), (dec, 0))
    tb.connect((dem, 0), (tap, 0))
    tb.msg_connect((dec, "out"), (dbg, "store"))
    sys.stdout.flush()
    saved = os.dup(1)
    with open(os.path.join(args.out, "stdout.txt"), "w+b") as cap:
        os.dup2(cap.fileno(), 1)
        try:
            tb.run()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    pdus = [bytes(pmt.u8vector_elements(pmt.cdr(dbg.get_message(i)))).hex() for i in range(dbg.num_messages())]
    llr = np.concatenate(tap.items) if tap.items else np.zeros(0, np.float32)
    llr.astype(np.float32).tofile(os.path.join(args.out, "llr.f32"))
    json.dump(tap.tags, open(os.path.join(args.out, "tags.json"), "w"))
    json.dump(pdus, open(os.path.join(args.out, "pdus.json"), "w"))
    print("dump_chain: %d items, %d tags, %d PDUs from %s" % (len(llr), len(tap.tags), len(pdus), ieee80211.__file__))


def load(d):
    return (np.fromfile(os.path.join(d, "llr.f32"), np.float32), json.load(open(os.path.join(d, "tags.json"))),
            json.load(open(os.path.join(d, "pdus.json"))))


def compare_dumps(ref_dir, our_dir):
    """{"ok": bool, "pdus_equal", "tags_equal", "llr_max_abs_diff", ...}: every finding by name, nothing raised"""
    rl, rt, rp = load(ref_dir)
    ol, ot, op = load(our_dir)
    res = {"reference_items": int(len(rl)), "our_items": int(len(ol)), "reference_pdus": len(rp), "our_pdus": len(op),
           "pdus_equal": rp == op, "llr_tolerance": LLR_TOL}
    own = [t for t in ot if t[1].startswith("c8p_")]
    ot = [t for t in ot if not t[1].startswith("c8p_")]
    res["our_own_tags_on_the_edge"] = sorted({t[1] for t in own})

    def by_packet(tags):
        d = {}
        for off, k, v in tags:
            d.setdefault(off, {})[k] = v
        return d
    rb, ob = by_packet(rt), by_packet(ot)
    bad, worst_f = [], 0.0
    if sorted(rb) != sorted(ob):
        bad.append({"what": "tag offsets differ", "only_reference": sorted(set(rb) - set(ob))[:5], "only_ours": sorted(set(ob) - set(rb))[:5]})
    for off in sorted(set(rb) & set(ob)):
        if set(rb[off]) != set(ob[off]):
            bad.append({"offset": off, "what": "tag keys differ", "reference": sorted(rb[off]), "ours": sorted(ob[off])})
            continue
        for k, v in rb[off].items():
            w = ob[off][k]
            if isinstance(v, float) and k in COMPUTED_FLOAT_KEYS:
                rel = abs(v - w) / max(abs(v), 1e-30)
                worst_f = max(worst_f, rel)
                if rel > 1e-4:
                    bad.append({"offset": off, "key": k, "reference": v, "ours": w})
            elif isinstance(v, list):
                a, b = np.array(v, np.float64), np.array(w, np.float64)
                if a.shape != b.shape or not np.allclose(a, b, rtol=1e-5, atol=1e-7):
                    bad.append({"offset": off, "key": k, "what": "c32vector differs"})
            elif v != w:
                bad.append({"offset": off, "key": k, "reference": v, "ours": w})
    res["tags_equal"] = not bad
    res["tag_differences"] = bad[:20]
    res["computed_float_tags_max_rel_diff"] = worst_f
    if len(rl) == len(ol):
        d = np.abs(rl.astype(np.float64) - ol.astype(np.float64))
        res["llr_max_abs_diff"] = float(d.max()) if len(d) else 0.0
        res["llr_mean_abs_diff"] = float(d.mean()) if len(d) else 0.0
        res["llr_items_over_tolerance"] = int((d > LLR_TOL).sum())
        res["llr_sign_flips"] = int(((rl > 0) != (ol > 0)).sum())
        if len(d) and d.max() > LLR_TOL:
            i = int(d.argmax())
            res["llr_worst_item"] = {"index": i, "reference": float(rl[i]), "ours": float(ol[i])}
    else:
        res["llr_max_abs_diff"] = None
    res["ok"] = bool(res["pdus_equal"] and res["tags_equal"] and res["llr_max_abs_diff"] is not None and res["llr_max_abs_diff"] <= LLR_TOL
                     and len(rp) > 0)
    return res


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--iq", required=True); r.add_argument("--iq2"); r.add_argument("--out", required=True); r.add_argument("--module-dir")
    c = sub.add_parser("compare")
    c.add_argument("reference"); c.add_argument("ours"); c.add_argument("--json")
    args = ap.parse_args()
    if args.cmd == "run":
        return run(args)
    res = compare_dumps(args.reference, args.ours)
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)
    print(json.dumps(res, indent=1))
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
