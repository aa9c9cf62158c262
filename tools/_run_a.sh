mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench.log 2>gpurun_out/final/bench.err; grep '^{' gpurun_out/final/bench.log > gpurun_out/final/bench_n1.json; python tools/bench_line.py final < gpurun_out/final/bench_n1.json | cut -c1-300
