# The driver's bench command under rocprofv3 --kernel-trace --stats -> profiles-ready summaries (run on the GPU box through gpurun).
#   bash tools/bench_rocprof.sh <tag> [extra bench.py arguments]
# writes gpurun_out/<tag>_bench_under_rocprof.json (the stdout line), <tag>_bench_detail_under_rocprof.json and <tag>_bench_kernel_stats.csv
tag=${1:-rX}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $R/gpurun_out/${tag}_bench_detail_under_rocprof.json "$@" > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/${tag}_bench_under_rocprof.err
cd $R
db=$(find /tmp/prof_$tag -name "*.db" | head -1)
python3 tools/rocprof_summary.py kernels "$db" gpurun_out/${tag}_bench_kernel_stats.csv 100
