"""gnuradio/ (SURVEY §8 f-3): the parts of the GNU Radio packaging that can be checked without GNU Radio.

* the stand-alone CMake configuration (-DC8P_STANDALONE=ON) configures and builds libc8p_blocks.so, and that library drives
  the mirrors exactly like the Makefile build;
* the block mirrors compile into namespace c8p (-DC8P_MIRROR_NS=c8p), which is how the GNU Radio module embeds them next to
  the public gr::ieee80211::demod ... classes (gr_complex etc. then come from <gnuradio/types.h>, so this check stops at the
  preprocessor);
* the GRC files keep the ids, make lines, parameter ids and ports the reference's flowgraphs rely on, the pybind files
  register the reference's Python names, the adapters create the reference's io signatures."""
import os
import re
import shutil
import subprocess

import pytest
import yaml

from c8p_testlib import ROOT

GR = os.path.join(ROOT, "gnuradio")


def test_standalone_cmake_configuration_builds_the_block_mirror(tmp_path):
    if shutil.which("cmake") is None:
        pytest.skip("cmake not installed")
    b = str(tmp_path / "build")
    subprocess.check_call(["cmake", "-S", GR, "-B", b, "-DC8P_STANDALONE=ON", "-DC8P_BUILD_ENGINE=OFF"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", b, "-j", "4"], stdout=subprocess.DEVNULL)
    lib = os.path.join(b, "libc8p_blocks.so")
    assert os.path.exists(lib)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
    for s in ("c8ph_create_ex", "c8ph_run", "c8ph_run_chain"):
        assert s in syms
    assert "N2gr9ieee802115demod4make" in syms                     # gr::ieee80211::demod::make: the reference's names


def test_engine_target_of_the_cmake_build_uses_the_makefile_flags():
    cm = open(os.path.join(GR, "CMakeLists.txt")).read()
    mk = open(os.path.join(ROOT, "gr-ieee80211_amd", "csrc", "Makefile")).read()
    for flag in ("-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt", "-disable-promote-alloca-to-lds=1"):
        assert flag in cm and flag in mk


def test_mirror_sources_compile_into_namespace_c8p_up_to_the_gnuradio_include():
    """-DC8P_MIRROR_NS=c8p: the only thing missing in this image is <gnuradio/types.h>"""
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DC8P_MIRROR_NS=c8p", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "gr-ieee80211_amd", "host", "c8p_blocks.cc")], capture_output=True, text=True)
    errs = [l for l in r.stderr.splitlines() if "error" in l]
    assert len(errs) >= 1 and all("gnuradio/types.h" in e for e in errs[:1]), r.stderr[:2000]


GRC = {
    "ieee80211_demod.block.yml": ("ieee80211_demod", "ieee80211.demod(${mupos}, ${mugid})", ["mupos", "mugid"], [("sig", "complex")], [("llr", "float")]),
    "ieee80211_demod2.block.yml": ("ieee80211_demod2", "ieee80211.demod2()", [], [("inSig1", "complex"), ("inSig2", "complex")], [("outLlr", "float")]),
    "ieee80211_decode.block.yml": ("ieee80211_decode", "ieee80211.decode(${ifdebug})", ["ifdebug"], [("inLlr", "float")], None),
}


@pytest.mark.parametrize("name", sorted(GRC))
def test_grc_files_keep_the_reference_contract(name):
    """ids / make lines / parameter ids / ports as grc/ieee80211_{demod,demod2,decode}.block.yml of the reference
    (ieee80211_demod.block.yml:5-32): examples/rx.grc and rx2.grc refer to exactly these"""
    bid, make, params, ins, outs = GRC[name]
    y = yaml.safe_load(open(os.path.join(GR, "grc", name)))
    assert y["id"] == bid and y["templates"]["make"] == make and y["templates"]["imports"] == "from gnuradio import ieee80211"
    assert [p["id"] for p in y.get("parameters", [])] == params
    assert [(i["label"], i["dtype"]) for i in y["inputs"]] == ins
    if outs is None:
        assert y["outputs"] == [{"domain": "message", "id": "out"}]
    else:
        assert [(o["label"], o["dtype"]) for o in y["outputs"]] == outs
    assert y["file_format"] == 1


def test_bindings_and_adapters_keep_names_and_signatures():
    bind = {b: open(os.path.join(GR, "python", "ieee80211", "bindings", b + "_python.cc")).read() for b in ("demod", "demod2", "decode")}
    assert 'm, "demod"' in bind["demod"] and 'py::arg("mupos"), py::arg("mugid")' in bind["demod"]
    assert 'm, "demod2"' in bind["demod2"] and 'm, "decode"' in bind["decode"] and 'py::arg("ifdebug")' in bind["decode"]
    mod = open(os.path.join(GR, "python", "ieee80211", "bindings", "python_bindings.cc")).read()
    assert "PYBIND11_MODULE(ieee80211_python, m)" in mod
    # the reference's ports (stream counts, item sizes) through c8p_stream_sig: the module's buffer type as the fourth argument of GNU Radio
    # 3.10's io_signature::make, or — safe mode — exactly the reference's three-argument call
    sig = {"demod": ("c8p_stream_sig(1, sizeof(gr_complex))", "c8p_stream_sig(1, sizeof(float))"),      # demod_impl.cc:36-37
           "demod2": ("c8p_stream_sig(2, sizeof(gr_complex))", "c8p_stream_sig(1, sizeof(float))"),     # demod2_impl.cc:36-37
           "decode": ("c8p_stream_sig(1, sizeof(float))", "gr::io_signature::make(0, 0, 0)")}           # decode_impl.cc:35-38
    for b, (i, o) in sig.items():
        src = open(os.path.join(GR, "lib", b + "_impl.cc")).read()
        assert i in src and o in src
        assert re.search(r"%s::sptr %s::make\(" % (b, b), src)
    ad0 = open(os.path.join(GR, "lib", "c8p_gr_adapter.h")).read()
    assert "gr::io_signature::make(n, n, (int)item, c8p_pinned_buffer::type)" in ad0 and "return gr::io_signature::make(n, n, (int)item);" in ad0
    assert 'message_port_register_out(pmt::mp("out"))' in open(os.path.join(GR, "lib", "decode_impl.cc")).read()
    ad = open(os.path.join(GR, "lib", "c8p_gr_adapter.h")).read()
    assert "TPP_DONT" in ad and 'pmt::mp("len")' in ad and "pmt::make_blob" in ad
    lib = open(os.path.join(GR, "lib", "CMakeLists.txt")).read()
    for kept in ("trigger_impl.cc", "sync_impl.cc", "signal_impl.cc", "cloud80211phy.cc", "modulation2_impl.cc", "pad2_impl.cc"):
        assert kept in lib
    assert "C8P_MIRROR_NS=c8p" in lib and 'find_package(Gnuradio "3.10"' in open(os.path.join(GR, "CMakeLists.txt")).read()


def test_adapters_type_check_against_the_gnu_radio_3_10_api_declarations():
    """gnuradio/lib/{demod,demod2,decode}_impl.cc + c8p_gr_adapter.h, and the block mirror in namespace c8p next to them, through the
    compiler: `g++ -fsyntax-only` against tests/gr_api_decls (DECLARATIONS of the GNU Radio 3.10 members the adapters use — see its
    README; GNU Radio itself is not in the image, gnuradio/ci/build.sh is the real build).  Catches misspelt members, wrong argument
    types, missing overrides and access errors (the scheduler calls are protected members of gr::block) here."""
    decls = os.path.join(ROOT, "tests", "gr_api_decls")
    inc = ["-I" + decls, "-I" + os.path.join(GR, "include"), "-I" + os.path.join(GR, "lib"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "gr-ieee80211_amd", "host")]
    srcs = [os.path.join(GR, "lib", f) for f in ("demod_impl.cc", "demod2_impl.cc", "decode_impl.cc")] + \
           [os.path.join(ROOT, "gr-ieee80211_amd", "host", "c8p_blocks.cc")]
    for extra in ([], ["-DC8P_GR_STOCK_API_ONLY"]):                # the full adapters, and the build that uses the stock gr::block API only
        for src in srcs:
            r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-DC8P_MIRROR_NS=c8p",
                                "-Dgnuradio_ieee80211_EXPORTS"] + extra + inc + [src], capture_output=True, text=True)
            assert r.returncode == 0, " ".join(extra) + " " + src + "\n" + r.stderr[-4000:]


def test_stock_api_only_build_touches_no_custom_buffer_topology_or_post_member():
    """-DC8P_GR_STOCK_API_ONLY: after the preprocessor, what is left of the adapters names none of the GNU Radio members the reference does not
    use itself — no buffer_single_mapped / buffer_type / block_detail / buffer_reader, no stream_buffer_type, no _post, no detail().  That is
    the build whose every GNU Radio call has a call site in the reference (c8p_gr_adapter.h's audit list), so it is the one to try first."""
    decls = os.path.join(ROOT, "tests", "gr_api_decls")
    inc = ["-I" + decls, "-I" + os.path.join(GR, "include"), "-I" + os.path.join(GR, "lib"), "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "gr-ieee80211_amd", "host")]
    for b in ("demod", "demod2", "decode"):
        r = subprocess.run(["g++", "-std=c++17", "-E", "-P", "-DC8P_MIRROR_NS=c8p", "-Dgnuradio_ieee80211_EXPORTS", "-DC8P_GR_STOCK_API_ONLY"] + inc +
                           [os.path.join(GR, "lib", b + "_impl.cc")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        text = r.stdout
        ours = text[text.index("c8p_gr_safe"):]                        # from the adapter on (what precedes it is GNU Radio's own headers / the decls)
        for banned in ("buffer_single_mapped", "c8p_pinned_buffer", "stream_buffer_type", "block_detail", "buffer_reader", "_post(", "detail()", "nreaders"):
            assert banned not in ours, (b, banned)
        assert "set_tag_propagation_policy" in ours and "get_tags_in_range" in ours and "message_port_pub" in ours


def test_side_by_side_comparison_of_two_edge_dumps(tmp_path):
    """gnuradio/ci/dump_chain.py compare — the half of test_reference_cxx_side_by_side that needs no GNU Radio: two dumps of the demod ->
    decode edge (items, tags, PDUs).  Equal dumps pass (this module's own c8p_* tags are reported, not compared); one LLR item off by
    2e-5, a differing integer tag, a missing PDU or a computed float tag off by 1e-3 each fail, by name."""
    import json
    import sys

    import numpy as np
    sys.path.insert(0, os.path.join(GR, "ci"))
    import dump_chain

    def write(d, llr, tags, pdus):
        os.makedirs(d, exist_ok=True)
        np.asarray(llr, np.float32).tofile(os.path.join(d, "llr.f32"))
        json.dump(tags, open(os.path.join(d, "tags.json"), "w"))
        json.dump(pdus, open(os.path.join(d, "pdus.json"), "w"))
    rng = np.random.default_rng(5)
    llr = rng.normal(0, 2, 4000).astype(np.float32)
    tags = [[0, "format", 2], [0, "len", 1458], [0, "cfo", 1234.5], [0, "sssnr0", 19.638828], [0, "total", 2000],
            [2000, "format", 0], [2000, "len", 100], [2000, "total", 2000], [2000, "mu2x1chan", [[1.0, 0.5], [0.25, -1.0]]]]
    pdus = ["0201", "00aa"]
    ref, ours = str(tmp_path / "ref"), str(tmp_path / "ours")
    write(ref, llr, tags, pdus)
    write(ours, llr + np.float32(4e-6) * np.sign(llr), tags + [[0, "c8p_llr", 77]], pdus)
    r = dump_chain.compare_dumps(ref, ours)
    assert r["ok"] and r["our_own_tags_on_the_edge"] == ["c8p_llr"] and 0 < r["llr_max_abs_diff"] <= 1e-5 and r["llr_sign_flips"] == 0
    bad = llr.copy(); bad[123] += 2e-5
    write(ours, bad, tags, pdus)
    r = dump_chain.compare_dumps(ref, ours)
    assert not r["ok"] and r["llr_items_over_tolerance"] == 1 and r["llr_worst_item"]["index"] == 123 and r["tags_equal"] and r["pdus_equal"]
    t2 = [list(t) for t in tags]; t2[1][2] = 1454
    write(ours, llr, t2, pdus)
    r = dump_chain.compare_dumps(ref, ours)
    assert not r["ok"] and not r["tags_equal"] and r["tag_differences"][0]["key"] == "len"
    t3 = [list(t) for t in tags]; t3[3][2] = 19.66
    write(ours, llr, t3, pdus)
    assert not dump_chain.compare_dumps(ref, ours)["tags_equal"]
    write(ours, llr, tags, pdus[:1])
    r = dump_chain.compare_dumps(ref, ours)
    assert not r["ok"] and not r["pdus_equal"]
    write(ours, llr[:-1], tags, pdus)
    assert dump_chain.compare_dumps(ref, ours)["llr_max_abs_diff"] is None
    # the command line: exit status 1 and a JSON record
    p = subprocess.run([sys.executable, os.path.join(GR, "ci", "dump_chain.py"), "compare", ref, ours, "--json", str(tmp_path / "r.json")], capture_output=True, text=True)
    assert p.returncode == 1 and json.load(open(tmp_path / "r.json"))["ok"] is False
