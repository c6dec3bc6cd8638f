"""gnuradio.ieee80211 — the gr-ieee80211 blocks; demod, demod2 and decode run on the MI355X engine (libc8p_rx.so)."""
import os

try:
    from .ieee80211_python import *      # noqa: F401,F403
except ModuleNotFoundError:
    pass
