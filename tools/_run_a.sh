set -u
out=gpurun_out/r3o
mkdir -p $out
export TMPDIR=/tmp
root=$PWD
for m in resnet_rpn swin_rpn swin_fcos; do
  timeout 300 python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-extras > $out/bench_$m.log 2>&1
  grep '^{' $out/bench_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'], d.get('final_loss'))"
done
for m in resnet_rpn swin_fcos; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$m -o p --output-format csv -- python $root/bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$out/prof_$m.log 2>&1)
f=$(find /tmp/prof_$m -name "*kernel_stats.csv" | head -1); cp $f $out/${m}_kernel_stats.csv
python - $out/${m}_kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print(sys.argv[1], 'total ms/step', tot/7e6)
for r in rows[:16]:
    print(f"  {r['Name'][:78]:78s} {int(r['Calls'])/7:6.1f} {float(r['TotalDurationNs'])/7e3:9.1f} us/step  avg {float(r['AverageNs'])/1e3:8.1f}")
PY
done
