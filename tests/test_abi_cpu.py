"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol include/c8p_rx.h
declares, agrees with abi.py on struct sizes, and its generated tables equal the reference's literals."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

from c8p_testlib import GOLD, ROOT, abi, oracle
from gr_ieee80211_amd import engine
from test_oracle_golden import FLT_TABLES, INT_TABLES


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "c8p_rx.h")).read()
    declared = set(re.findall(r"\b(c8p_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(engine.SYMBOLS)
    lib = engine.load_library()
    for s in declared:
        assert hasattr(lib, s), s


def test_engine_refuses_to_run_without_a_gpu_or_runs_on_one():
    import torch
    if torch.cuda.is_available():
        engine.Engine().close()
    else:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            engine.Engine()


@pytest.mark.parametrize("name", sorted(INT_TABLES))
def test_engine_int_tables_match_reference_literals(name):
    which, sub = INT_TABLES[name]
    t = engine.table(which, sub).astype(np.int32)
    d = json.load(open(os.path.join(GOLD, "table_digests.json")))[name]
    assert len(t) == d["n"] and hashlib.sha256(t.tobytes()).hexdigest() == d["sha256"]
    assert (t == oracle().table(which, sub)).all()


@pytest.mark.parametrize("name", sorted(FLT_TABLES))
def test_engine_float_tables_match_reference_literals(name):
    t = engine.table(FLT_TABLES[name], 0).astype(np.float32)
    d = json.load(open(os.path.join(GOLD, "table_digests.json")))[name]
    assert len(t) == d["n"] and hashlib.sha256(t.tobytes()).hexdigest() == d["sha256"]


def test_product_never_touches_the_oracle():
    """The shipped path (package + include) must not import, link or name anything under oracle/."""
    bad = []
    for base in ("gr-ieee80211_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".cc", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle|c8po_", txt):
                        bad.append(os.path.join(dp, f))
    assert bad == []
