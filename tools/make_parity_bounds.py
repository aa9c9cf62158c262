#!/usr/bin/env python
"""tests/golden/parity_bounds.json from a measurement log (NRPN_PARITY_LOG of a full `pytest -m gpu` run on an MI355X):
bound = 2 x the measured worst error of the case, floored at the comparison's own rounding noise (tests/parity_log.py FLOOR).
    python tools/make_parity_bounds.py gpurun_out/r5a/parity_measured.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from parity_log import FLOOR  # noqa: E402

src = sys.argv[1]
measured = json.load(open(src))
bounds = {k: max(2.0 * v, FLOOR[k.rsplit("/", 1)[1]]) for k, v in sorted(measured.items())}
json.dump(bounds, open(os.path.join(ROOT, "tests", "golden", "parity_bounds.json"), "w"), indent=1, sort_keys=True)
json.dump({"source": "NRPN_PARITY_LOG of python -m pytest tests -m gpu on one MI355X (tools/r5_call1.sh)", "measured": measured},
          open(os.path.join(ROOT, "profiles", "r05_parity_measured.json"), "w"), indent=1, sort_keys=True)
print(f"{len(bounds)} bounds written")
