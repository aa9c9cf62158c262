#!/bin/bash
# PMC counters of the dominant conv launches (tools/one_conv.py: 256->256, 3^3, 40^3, bf16), one rocprofv3 --pmc pass per
# counter group (MI355X_MICROARCH.md: never mix --pmc with the trace domains other than --kernel-trace).  Run on the GPU box:
#   bash tools/pmc_conv.sh gpurun_out/pmc   ->  gpurun_out/pmc/pmc_summary.json
set -u
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
export TMPDIR=/tmp
root=$(cd "$(dirname "$0")/.." && pwd)
i=0
groups=${PMC_GROUPS:-all}
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i + 1))
  if [ "$groups" != "all" ] && [[ ",$groups," != *",$i,"* ]]; then continue; fi
  (cd /tmp && timeout 170 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_pass$i -- python "$root/tools/one_conv.py" > /tmp/pmc_pass$i.log 2>&1)
  f=$(ls /tmp/pmc_pass$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" "$out/pass$i.csv"
done
NRPN_ROOT="$root" python - "$out" <<'PY'
import csv, json, sys, collections, glob, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(float))     # (kernel, dispatch) -> counter -> value
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), cs in per.items():
        if "conv_" not in k:
            continue
        for c, v in cs.items():
            acc[k.split("(")[0].replace("void ", "")][c].append(v)
res = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in d:      # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads: x2
        d["hbm_read_bytes"] = d["FETCH_SIZE"] * 2 * 1024
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024 if "WRITE_SIZE" in d else None
        d["hbm_bytes"] = d["hbm_read_bytes"] + (d["hbm_write_bytes"] or 0) if "WRITE_SIZE" in d else None
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    res[k] = d
import hashlib
sha = hashlib.sha256()
root = os.path.join(os.path.dirname(os.path.abspath(out)), "..") if False else os.environ.get("NRPN_ROOT", ".")
for f in ("conv3d.hip", "conv_halo.hip", "conv_common.cuh", "common.h"):      # the same hash bench.py computes (conv_source_hash): counters are tied to the kernel source
    sha.update(open(os.path.join(root, "nerf_rpn_amd", "csrc", f), "rb").read())
json.dump({"conv_source_sha16": sha.hexdigest()[:16], "method": "rocprofv3 --pmc, one pass per counter group (FETCH_SIZE and WRITE_SIZE in separate passes: 3 + 2 of the 4 TCC slots) on "
                     "tools/one_conv.py: 256->256 k3 @40^3 bf16, per-launch averages; bytes = FETCH_SIZE[KiB] x 1024 x 2 (gfx950 tallies the "
                     "128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md 'HBM') + WRITE_SIZE[KiB] x 1024 (uncalibrated)",
           "kernels": res}, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:2500])
PY
for f in "$out"/pass*.csv; do rm -f "$f"; done
