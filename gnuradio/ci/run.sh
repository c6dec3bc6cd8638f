#!/bin/sh
# gnuradio/ci/run.sh — the flowgraph contract test of SURVEY.md §8 f-3, on a machine with GNU Radio 3.10, the module of gnuradio/
# installed (gnuradio/ci/build.sh) and an MI355X:   one command, from the repository root.  Writes GRCI.json beside this script's
# repository root: versions, and pass / fail / skip per test — the record to attach to the repository once somebody has run it.
set -u
HERE=$(cd "$(dirname "$0")/../.." && pwd)
cd "$HERE"
rm -f "$HERE/gnuradio/ci/side_by_side.json"
export C8P_RX_LIB="$HERE/gr-ieee80211_amd/libc8p_rx.so"
python3 -m pytest tests/test_gnuradio_flowgraph.py tests/test_gnuradio_packaging.py -m "gpu or not gpu" -q -rs --junitxml="$HERE/gnuradio/ci/junit.xml"
RC=$?
python3 - "$HERE" "$RC" <<'P'
import json, platform, subprocess, sys, time, xml.etree.ElementTree as ET
root, rc = sys.argv[1], int(sys.argv[2])


def out(cmd):
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[0]
    except Exception as e:
        return "unavailable (%s)" % type(e).__name__


tests = []
try:
    for tc in ET.parse(root + "/gnuradio/ci/junit.xml").getroot().iter("testcase"):
        kind = "passed"
        for child in tc:
            if child.tag in ("failure", "error"):
                kind = "failed"
            elif child.tag == "skipped":
                kind = "skipped: " + (child.get("message") or "")[:160]
        tests.append({"test": tc.get("classname", "") + "::" + tc.get("name", ""), "result": kind})
except Exception as e:
    tests.append({"test": "junit.xml", "result": "unreadable: %s" % e})
rec = {"when": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "pytest_exit_code": rc,
       "gnuradio": out(["gnuradio-config-info", "--version"]), "python": platform.python_version(),
       "rocm": out(["cat", "/opt/rocm/.info/version"]), "gpu": out(["rocminfo"]) if False else out(["sh", "-c", "rocminfo | grep -m1 gfx"]),
       "head": out(["git", "-C", root, "rev-parse", "HEAD"]),
       "passed": sum(t["result"] == "passed" for t in tests), "failed": sum(t["result"] == "failed" for t in tests),
       "skipped": sum(t["result"].startswith("skipped") for t in tests), "tests": tests}
try:     # the reference's own C++ blocks beside ours on the same IQ (test_reference_cxx_side_by_side): max |LLR difference|, tags, PDUs
    rec["reference_cxx_side_by_side"] = json.load(open(root + "/gnuradio/ci/side_by_side.json"))
except Exception:
    rec["reference_cxx_side_by_side"] = "not run (gnuradio/ci/build_reference.sh <checkout> first)"
json.dump(rec, open(root + "/GRCI.json", "w"), indent=1)
print("GRCI.json: %d passed, %d failed, %d skipped (GNU Radio %s)" % (rec["passed"], rec["failed"], rec["skipped"], rec["gnuradio"]))
P
exit $RC
