mkdir -p gpurun_out/r3p
timeout 300 python tools/power_probe.py > gpurun_out/r3p/power.log 2>&1; cat gpurun_out/r3p/power.log | grep -v Warning | tail -40
