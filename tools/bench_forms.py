#!/usr/bin/env python
"""A/B helper for a GPU visit: bench.py with comparison forms of mnk.knobs.FORMS flipped (they are not environment switches of the
product).  Usage: python tools/bench_forms.py DGRAD_BN_STATS=0[,NAME=1...] [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "monkey-net_amd"))
from mnk import knobs  # noqa: E402

spec = sys.argv[1]
for item in filter(None, spec.split(",")):
    name, value = item.split("=")
    assert name in knobs.FORMS, name
    knobs.FORMS[name] = value not in ("0", "false", "False")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
