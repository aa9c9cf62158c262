set -u
mkdir -p gpurun_out/r3s
root=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_postproc.py tests/test_gpu_stages.py -x -q 2>&1 | tail -3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/profe -o p --output-format csv -- python $root/tools/eval_protocol.py 10 > $root/gpurun_out/r3s/prof.log 2>&1)
f=$(find /tmp/profe -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r3s/eval_kernel_stats.csv
grep -E "nms_" gpurun_out/r3s/eval_kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,100-180
tail -4 gpurun_out/r3s/prof.log
