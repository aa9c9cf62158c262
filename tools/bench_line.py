"""Pull the fields of interest out of bench.py's JSON line on stdin:  python bench.py ... | python tools/bench_line.py <label>"""
import json
import sys

for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {"frac": None}
        print(sys.argv[1] if len(sys.argv) > 1 else "", "ms/step", d["ms_per_step"], "frac", r["frac"], "fwd", (r.get("forward_vgg19_fpn") or {}).get("ms"),
              [(c["call"][7:], c["voxels"], c["cin"], c["cout"], c["avg_us"]) for c in (d.get("conv_breakdown") or r.get("conv_breakdown") or [])[:6]], flush=True)
