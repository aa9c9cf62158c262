"""Measured parity errors of the end-to-end fixtures: every e2e test records the WORST box / score / feature / loss error it
saw against the reference's golden vectors, prints it, and -- when tests/golden/parity_bounds.json holds an entry for the case -- asserts it
against that entry (= 2 x the value measured on an MI355X when the bound was taken, profiles/r05_parity_measured.json) instead of only against
the blanket tolerance.  NRPN_PARITY_LOG=<path> appends the measurements to a JSON file (how the bounds were collected)."""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_BOUNDS_PATH = os.path.join(_HERE, "golden", "parity_bounds.json")
_bounds = None
FLOOR = {"box": 2e-5, "score": 2e-7, "feat": 2e-6, "loss": 2e-6}      # below this a doubled measurement is rounding noise of the comparison itself


def bounds():
    global _bounds
    if _bounds is None:
        _bounds = json.load(open(_BOUNDS_PATH)) if os.path.exists(_BOUNDS_PATH) else {}
    return _bounds


def record(case, kind, value, blanket=None):
    """case: e.g. 'eval_obb_s2[0]/fp32'; kind: box | score | feat | loss.  Returns the bound that was applied (None if only the blanket one)."""
    value = float(value)
    print(f"[parity] {case} {kind}: measured {value:.3e}" + (f" (blanket tolerance {blanket:.1e})" if blanket is not None else ""))
    path = os.environ.get("NRPN_PARITY_LOG")
    if path:
        try:
            d = json.load(open(path)) if os.path.exists(path) else {}
        except Exception:
            d = {}
        key = f"{case}/{kind}"
        d[key] = max(value, d.get(key, 0.0))
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    b = bounds().get(f"{case}/{kind}")
    if b is not None:
        assert value <= b, f"{case} {kind}: {value:.3e} exceeds the measured-x2 bound {b:.3e} (tests/golden/parity_bounds.json)"
    return b
