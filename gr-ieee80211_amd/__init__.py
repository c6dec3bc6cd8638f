"""MI355X-native 802.11 OFDM RX hot path (demod / demod2 / decode) behind a C ABI.

The directory name keeps the upstream project's hyphen; load it under the module name
`gr_ieee80211_amd` with `__graft_entry__.load_package()`.
"""
from . import abi  # noqa: F401
