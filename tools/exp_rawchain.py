#!/usr/bin/env python3
"""tools/exp_rawchain.py — where the time goes when the chain starts at the RAW stream (rx.grc from the SDR source on):
one 33.5 M-item stream of cfg-2 packets resident in HBM -> c8p_detect -> c8p_signal -> (signal's consumption rule, host)
-> c8p_rx -> PDUs.  Prints the wall time of every stage over K repetitions and the chain's packets/s and samples/s.
Run on the GPU box.  C8P_DET_TLOG=1 adds c8p_detect's own laps on stderr."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from c8p_cases import dewindow
from c8p_testlib import abi, load_npz, through_channel
from gr_ieee80211_amd import engine
from test_detect_frontend import consume_filter


def main(n_target=32 << 20, K=6):
    dev = torch.device("cuda:0")
    d = load_npz("packets_cfg.npz")["cfg2_VHT_mcs7_1458B_0"]
    rng = np.random.default_rng(1)
    unit = through_channel(dewindow(d["tx"])[:1], None, 30.0, rng, lead=600)[0]
    reps = max(1, n_target // len(unit))
    x = torch.from_numpy(unit.view(np.float32)).to(dev).repeat(reps)
    n = x.numel() // 2
    arr = engine.DevArray(x.data_ptr(), n, x)
    e_det, e_sig, e_rx = engine.Engine(device=0), engine.Engine(device=0, profile=True), engine.Engine(device=0, profile=True)
    laps = {k: [] for k in ("detect", "sig_in", "signal", "fetch_pkt", "filter", "rx", "outs", "total")}
    for it in range(K + 1):
        t = [time.perf_counter()]
        ev = e_det.detect(arr); t.append(time.perf_counter())
        s = np.zeros(len(ev), abi.SIG_IN_DT)
        s["raw_off"][:, 0] = ev["index"]; s["n_avail"] = np.minimum(n - ev["index"], 224 + 109280)     # longest legacy packet (4095 B at 6 Mb/s)
        s["cfo_rad"], s["snr"], s["rssi"], s["n_ant"] = ev["rad"], ev["snr"], ev["rssi"], 1
        t.append(time.perf_counter())
        gpk = e_sig.signal(s, arr, consume=True); e_sig.sync(); t.append(time.perf_counter())
        t.append(time.perf_counter())
        keep = np.flatnonzero(gpk["n_items"]); t.append(time.perf_counter())
        pi, bi = e_sig.buffer(abi.BUF_SIG_IQ); pc, bc = e_sig.buffer(abi.BUF_SIG_CHAN)
        e_rx.rx(gpk[keep], engine.DevArray(pi, bi // 8), engine.DevArray(pc, bc // 8)); e_rx.sync(); t.append(time.perf_counter())
        outs = e_rx.outs(); t.append(time.perf_counter())
        if it:
            for k, a, b in zip(list(laps)[:-1], t[:-1], t[1:]):
                laps[k].append(1e3 * (b - a))
            laps["total"].append(1e3 * (t[-1] - t[0]))
    kt = {k: round(ms / cnt, 4) for e in (e_sig, e_rx) for k, (ms, cnt) in e.kernel_times().items() if cnt}
    ok = int((outs["n_pdu"] == 1).sum())
    med = {k: round(float(np.median(v)), 4) for k, v in laps.items()}
    print(json.dumps({"samples": n, "packets": reps, "events": int(len(ev)), "kept": len(keep), "pdus_ok": ok, "ms": med, "kernel_ms": kt,
                      "packets_per_s": round(ok / (med["total"] * 1e-3)), "samples_per_s": round(n / (med["total"] * 1e-3))}))


if __name__ == "__main__":
    main()
