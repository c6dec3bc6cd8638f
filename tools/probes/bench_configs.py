#!/usr/bin/env python3
"""tools/probes/bench_configs.py — c8p_rx throughput on the OTHER BASELINE.json configs (the bench line is configs[1] only):
cfg3 HT MCS15 2x2 x1024, cfg4 VHT MCS8 x8192, cfg5 MU-MIMO 1024 per station, cfg1 legacy MCS0 x4096.  Batches are noisy
copies of the committed reference-TX fixtures, resident in HBM; every published PDU is checked.  Prints one dict per config.
Run on the GPU box: python tools/probes/bench_configs.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    from c8p_cases import replicated_batch
    from c8p_testlib import GOLD, H22, load_npz
    from gr_ieee80211_amd import engine
    cfg = load_npz("packets_cfg.npz")
    mu = np.load(os.path.join(GOLD, "packets_mu.npz"))
    cases = [
        ("cfg1 legacy MCS0 100 B x4096", cfg["cfg1_L_mcs0_100B"], cfg["cfg1_L_mcs0_100B"]["pdu"].tobytes(), 4096, 1, None, 0, 30.0),
        ("cfg3 HT MCS15 2x2 1458 B x1024", cfg["cfg3_HT_mcs15_2x2_1458B"], cfg["cfg3_HT_mcs15_2x2_1458B"]["pdu"].tobytes(), 1024, 2, H22, 0, 30.0),
        ("cfg3 HT MCS15 2x2 1458 B x4096", cfg["cfg3_HT_mcs15_2x2_1458B"], cfg["cfg3_HT_mcs15_2x2_1458B"]["pdu"].tobytes(), 4096, 2, H22, 0, 30.0),
        ("cfg2 VHT MCS7 1458 B x4096 (fixture frame)", cfg["cfg2_VHT_mcs7_1458B_0"], cfg["cfg2_VHT_mcs7_1458B_0"]["pdu"].tobytes(), 4096, 1, None, 0, 30.0),
        ("cfg4 VHT MCS8 1458 B x8192", cfg["cfg4_VHT_mcs8_1458B"], cfg["cfg4_VHT_mcs8_1458B"]["pdu"].tobytes(), 8192, 1, None, 0, 36.0),
        ("cfg5 MU-MIMO station 0 (MCS3 100 B) x1024", {"tx": mu["mu_tx"]}, mu["pdu0"].tobytes(), 1024, 1, mu["h"][0:1], 0, 30.0),
        ("cfg5 MU-MIMO station 1 (MCS3 60 B) x1024", {"tx": mu["mu_tx"]}, mu["pdu1"].tobytes(), 1024, 1, mu["h"][1:2], 1, 30.0),
    ]
    eng = engine.Engine(device=0)
    for name, frame, want, n, n_ant, H, mupos, snr in cases:
        b = replicated_batch(frame, n, n_ant, snr_db=snr, seed=31 + n, H=H, mupos=mupos)
        pk, iq, chan = b.arrays()
        iq_d = torch.from_numpy(iq.view(np.float32)).to(dev)
        ch_d = torch.from_numpy(chan.view(np.float32)).to(dev)
        A, C = engine.DevArray(iq_d.data_ptr(), iq.size, iq_d), engine.DevArray(ch_d.data_ptr(), chan.size, ch_d)
        for _ in range(3):
            eng.rx(pk, A, C)
        res = eng.results(want_llr=False)
        ok = sum(1 for r in res if r["pdus"] == [want])
        assert all(r["pdus"] in ([], [want]) for r in res), name
        steps = 30
        eng.sync()
        t = time.perf_counter()
        for _ in range(steps):
            eng.rx(pk, A, C)
        eng.sync()
        dt = time.perf_counter() - t
        print({"config": name, "packets_per_s": round(n * steps / dt), "ms_per_step": round(1e3 * dt / steps, 4), "fcs_ok": ok / n})
    eng.close()


if __name__ == "__main__":
    main()
