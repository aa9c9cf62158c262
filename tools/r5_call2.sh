#!/bin/bash
# GPU call 2 of round 5: the tests that failed / changed in call 1, PMC of the halo kernel's two K orders, kernel-stats runs, the bench line.
cd "$(dirname "$0")/.."
O=gpurun_out/r5b
mkdir -p $O
root=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_trainer.py tests/test_gpu_graph.py -q -m gpu -k "bf16x3 or halo or split or batchnorm or trains_like or second_trainer" > $O/t_a.log 2>&1; tail -6 $O/t_a.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -m gpu -k "bf16x3 or deviates_like or vgg_af or swin_t or swin_l" > $O/t_b.log 2>&1; tail -6 $O/t_b.log
timeout 300 bash tools/pmc_conv.sh $O/pmc > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-300
timeout 400 python bench.py --steps 100 --warmup 10 > $O/bench.log 2> $O/bench.err; grep '^{' $O/bench.log > $O/bench_n1.json; python tools/bench_line.py r5b < $O/bench_n1.json | cut -c1-300
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench2.log 2>&1)
cp $(find /tmp/prof2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
(cd /tmp && NRPN_WGRAD_STREAM=0 timeout 90 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe --no-extras > $root/$O/prof_bench.log 2>&1)
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single_stream.csv
python tools/prof_summary.py $(find /tmp/prof -name "*kernel_trace.csv" | head -1) $O/kernel_summary_single_stream.json 7
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
