# rocprofv3 passes whose summaries are committed under profiles/ (run on the GPU box through gpurun):
#   bash tools/bench_rocprof.sh <tag> bench [bench.py arguments]   the driver's bench command under --kernel-trace --stats
#        -> gpurun_out/<tag>_bench_under_rocprof.json (the stdout line), <tag>_bench_detail_under_rocprof.json, <tag>_bench_kernel_stats.csv
#   bash tools/bench_rocprof.sh <tag> timeline [f32s|bf16]         kernel trace of configs[1] searched to completion -> per-step timelines
#        at four points of the launch-size ladder -> gpurun_out/<tag>_step_timelines_<dtype>.txt, <tag>_solve_run_<dtype>.json
#   bash tools/bench_rocprof.sh <tag> select_pmc                   SQ instruction / wait counters of the tree kernel in the steady-state pool
#        -> gpurun_out/<tag>_select_pmc.txt
#   bash tools/bench_rocprof.sh <tag> env_traffic                  FETCH_SIZE / WRITE_SIZE passes (separate) of the environment kernels at 2^24
#        states -> gpurun_out/<tag>_env_pmc_traffic.json
#   bash tools/bench_rocprof.sh <tag> env_stats                    kernel durations of the environment kernels at 2^24 states (tools/env_bench.py 24 under
#        --kernel-trace --stats) -> gpurun_out/<tag>_env_kernel_stats.csv
#   bash tools/bench_rocprof.sh <tag> gemm_pmc                     matrix-pipe duty cycle of the hidden-layer kernel: SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
#        wave / wait / LDS counters over tools/split_gemm_fused_probe.py at 11 264 rows (the program itself behind `--`)
#        -> gpurun_out/<tag>_split_gemm_pmc.txt
tag=${1:-rX}; mode=${2:-bench}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
case $mode in
bench)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --detail $O/${tag}_bench_detail_under_rocprof.json "$@" > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_bench_under_rocprof.err
  python3 $R/tools/rocprof_summary.py kernels "$(find /tmp/prof_$tag -name '*.db' | head -1)" $O/${tag}_bench_kernel_stats.csv 100 ;;
timeline)
  dt=${1:-f32s}
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl_$tag -- python3 $R/tools/search_probe.py solve --dtype $dt --out $O/${tag}_solve_run_$dt.json > $O/${tag}_timeline.log 2>&1
  f=$(find /tmp/prof_tl_$tag -name '*kernel_trace.csv' | head -1)
  for s in -400 -1500 -3000 -4500; do echo "=== steps from $s"; python3 $R/tools/rocprof_summary.py timeline $f $s 150 1; done > $O/${tag}_step_timelines_$dt.txt ;;
select_pmc)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d /tmp/prof_pmc_$tag -- python3 $R/tools/search_probe.py window bf16 20 > $O/${tag}_select_pmc.log 2>&1
  python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_pmc_$tag -name '*counter_collection.csv' | head -1)" k_mcts_select > $O/${tag}_select_pmc.txt ;;
gemm_pmc)
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/prof_gp_$tag -- python3 $R/tools/split_gemm_fused_probe.py --rows 11264 --shapes 4096x2048 --reps 8 > $O/${tag}_split_gemm_pmc.log 2>&1
  python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_gp_$tag -name '*counter_collection.csv' | head -1)" k_split_gemm > $O/${tag}_split_gemm_pmc.txt
  rocprofv3 --kernel-trace --stats -d /tmp/prof_gs_$tag -o g -- python3 $R/tools/split_gemm_fused_probe.py --rows 11264 --shapes 4096x2048 --reps 8 >> $O/${tag}_split_gemm_pmc.log 2>&1
  python3 $R/tools/rocprof_summary.py kernels "$(find /tmp/prof_gs_$tag -name '*.db' | head -1)" $O/${tag}_split_gemm_kernel_stats.csv 6
  tail -3 $O/${tag}_split_gemm_pmc.log >> $O/${tag}_split_gemm_pmc.txt ;;
gemm_lds)
  # LDS / matrix-pipe counters of the hidden-layer kernel at 196 608 rows, this build and (RUBIKS_HIP_LIB_B) a diagnostic build of it
  for v in tree ${RUBIKS_HIP_LIB_B:+b}; do
    if [ $v = b ]; then export RUBIKS_HIP_LIB=$RUBIKS_HIP_LIB_B; fi
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/prof_gl_${tag}_$v -- python3 $R/tools/gemm_tile_ab.py --tiles 1 --rows 196608 --reps 4 > $O/${tag}_gemm_lds_$v.log 2>&1
    echo "== build: ${RUBIKS_HIP_LIB:-tree}"; python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_gl_${tag}_$v -name '*counter_collection.csv' | head -1)" k_split_gemm
  done > $O/${tag}_gemm_lds_pmc.txt ;;
gemm_traffic)
  # HBM bytes per launch of the hidden-layer kernel at the step's 11 264 rows: FETCH_SIZE and WRITE_SIZE in separate passes
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_gf_$tag -- python3 $R/tools/gemm_tile_ab.py --tiles 1 --rows 11264 --reps 8 > $O/${tag}_gemm_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_gw_$tag -- python3 $R/tools/gemm_tile_ab.py --tiles 1 --rows 11264 --reps 8 > $O/${tag}_gemm_write.log 2>&1
  { echo "FETCH_SIZE pass"; python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_gf_$tag -name '*counter_collection.csv' | head -1)" k_split_gemm
    echo "WRITE_SIZE pass"; python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_gw_$tag -name '*counter_collection.csv' | head -1)" k_split_gemm; } > $O/${tag}_gemm_traffic.txt ;;
env_stats)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_es_$tag -o env -- python3 $R/tools/env_bench.py 24 > $O/${tag}_env_bench.log 2>&1
  python3 $R/tools/rocprof_summary.py kernels "$(find /tmp/prof_es_$tag -name '*.db' | head -1)" $O/${tag}_env_kernel_stats.csv 12 ;;
env_traffic)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f_$tag -- python3 $R/tools/env_bench.py 24 > $O/${tag}_env_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w_$tag -- python3 $R/tools/env_bench.py 24 > $O/${tag}_env_write.log 2>&1
  python3 $R/tools/rocprof_summary.py traffic "$(find /tmp/prof_f_$tag -name '*counter_collection.csv' | head -1)" "$(find /tmp/prof_w_$tag -name '*counter_collection.csv' | head -1)" $O/${tag}_env_pmc_traffic.json 24 ;;
esac
