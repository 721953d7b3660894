#!/bin/bash
# A/B of the tree kernel's re-validation on the GPU box: builds each variant, runs the steady-state pool under
# rocprofv3 (kernel trace, then FETCH_SIZE, then WRITE_SIZE in their own passes) and prints one summary per variant.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/select_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
variant() {   # name, EXTRA flags
  touch $R/rl-rubiks_amd/csrc/rubiks_mcts.hip
  make -C $R/rl-rubiks_amd EXTRA="$2" > $OUT/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $OUT/build_$1.log; return; }
  echo "== $1 ($2)"
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$1 -o t -- python3 $R/tools/window_probe.py bf16 20 > $OUT/probe_$1.log 2>&1
  grep window $OUT/probe_$1.log | tail -4
  python3 $R/tools/select_ab.py trace $(find $OUT/t_$1 -name "*kernel_trace.csv" | head -1)
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/p_$1_$C -o p -- python3 $R/tools/window_probe.py bf16 20 > /dev/null 2>&1
    python3 $R/tools/select_ab.py pmc $(find $OUT/p_$1_$C -name "*counter_collection.csv" | head -1)
  done
  rm -rf $OUT/t_$1 $OUT/p_$1_*
}
variant two_lines "-DRC_SELECT_ONE_LINE=0"
variant one_line "-DRC_SELECT_ONE_LINE=1"
variant one_line_pf0 "-DRC_SELECT_ONE_LINE=1 -DRC_SELECT_PF_LINE1=0"
touch $R/rl-rubiks_amd/csrc/rubiks_mcts.hip
make -C $R/rl-rubiks_amd EXTRA="-DRC_SELECT_FASTSTATS" > $OUT/build_stats.log 2>&1 && python3 $R/tools/select_ab.py faststats
