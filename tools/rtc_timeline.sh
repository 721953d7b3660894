# Kernel trace of configs[1] searched to completion (fp32-accurate engine) -> per-step timelines at four points of the launch-size ladder.
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python3 /root/repo/tools/solve_run_profile.py --max-states 175000 --dtype ${DTYPE:-f32s} --no-sizes --out /root/repo/gpurun_out/r3g_solve_run_${DTYPE:-f32s}.json > /root/repo/gpurun_out/r3g_rtc.log 2>&1
cd /root/repo
f=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
for s in -400 -1500 -3000 -4500; do echo "=== steps from $s"; python3 tools/step_timeline.py $f $s 150; done > gpurun_out/r3g_step_timelines_${DTYPE:-f32s}.txt
