#!/bin/bash
# NEEDS tools/patches/r6_fused_splitk.patch applied (git apply) and the library rebuilt (the fused split-K experiment is not merged).
# round 6, call 4: fused split-K (20^3-class launches on the 256x256 tile): parity + forward sequence + bench A/B against the two-launch form
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-r6c4}
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_conv.py -q -m gpu -p no:cacheprovider -x -k "fused_split or large_tile or deterministic" > $O/t_fused.log 2>&1; tail -5 $O/t_fused.log
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/fwd -o p --output-format csv -- python $root/tools/forward_trace.py 20 > $root/$O/forward_prof.log 2>&1)
cp $(find /tmp/fwd -name "*kernel_stats.csv" | head -1) $O/forward_kernel_stats.csv
python tools/forward_trace.py --seq $(find /tmp/fwd -name "*kernel_trace.csv" | head -1) 20 > $O/forward_seq.txt; tail -1 $O/forward_seq.txt
for i in 1 2; do
NRPN_FUSED_SPLIT=0 python tools/forward_trace.py 50 2>&1 | tail -1 | sed 's/^/two-launch: /'
python tools/forward_trace.py 50 2>&1 | tail -1 | sed 's/^/fused:      /'
done
for f in 0 1; do
NRPN_FUSED_SPLIT=$f timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_f$f.log 2> $O/bench_f$f.err; grep '^{' $O/bench_f$f.log > $O/bench_f$f.json
python tools/bench_line.py fused=$f < $O/bench_f$f.json | cut -c1-300
done
