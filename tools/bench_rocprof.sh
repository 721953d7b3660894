# The driver's bench command under rocprofv3 --kernel-trace --stats -> profiles-ready summaries (run on the GPU box through gpurun).
#   bash tools/bench_rocprof.sh <tag>      writes gpurun_out/<tag>_bench_under_rocprof.json and gpurun_out/<tag>_bench_kernel_stats.csv
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python3 /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /root/repo/gpurun_out/${tag}_bench_under_rocprof.json 2> /root/repo/gpurun_out/${tag}_bench_under_rocprof.err
cd /root/repo
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 tools/kernel_stats.py "$db" gpurun_out/${tag}_bench_kernel_stats_full.csv && head -101 gpurun_out/${tag}_bench_kernel_stats_full.csv > gpurun_out/${tag}_bench_kernel_stats.csv && rm gpurun_out/${tag}_bench_kernel_stats_full.csv
