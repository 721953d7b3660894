#!/bin/bash
# input layer of the split engine: the two waves of a SIMD in phase (RC_FL_PINGPONG=0) / alternating between matrix stage and epilogue (1)
R=${GRAFT_REPO_ROOT:-/root/repo}
for PP in 0 1; do
  touch $R/rl-rubiks_amd/csrc/rubiks_net.hip
  make -C $R/rl-rubiks_amd EXTRA="-DRC_FL_PINGPONG=$PP" > /tmp/build_pp_$PP.log 2>&1 || { echo "build PP=$PP failed"; tail -5 /tmp/build_pp_$PP.log; continue; }
  echo "== RC_FL_PINGPONG=$PP"; timeout -k 10 120 python3 $R/tools/first_layer_split_bench.py --quick 2>&1 | grep -v amdgpu.ids
done
