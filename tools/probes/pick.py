"""tools/probes/pick.py <label> — reads bench.py's JSON line from stdin, prints packets/s, ms per step and the exclusive forward / finish times"""
import json, sys
for line in sys.stdin:
    if line.startswith('{"metric"'):
        j = json.loads(line); x = j["roofline"]
        print(sys.argv[1], round(j["value"]), round(j["ms_per_step"], 4), "excl", x["kernel"], x["launch_ms"], "fcs", j["config"].get("fcs_ok_fraction"))
