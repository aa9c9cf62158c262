set -u
out=gpurun_out/r3n
mkdir -p $out
root=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_stages.py -x -q 2>&1 | tail -4
timeout 300 python tools/eval_protocol.py 20 > $out/eval.log 2>&1; tail -4 $out/eval.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profe -o p --output-format csv -- python $root/tools/eval_protocol.py 10 > $root/$out/prof.log 2>&1)
f=$(find /tmp/profe -name "*kernel_stats.csv" | head -1); cp $f $out/eval_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3n/eval_kernel_stats.csv')))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls'])/13:6.1f} {float(r['TotalDurationNs'])/13e3:9.1f} us/fwd  avg {float(r['AverageNs'])/1e3:8.1f}")
PY
