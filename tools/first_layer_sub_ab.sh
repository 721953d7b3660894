#!/bin/bash
# input layer of the split engine with 2 / 3 / 4 state tiles per wave (RC_FL_SUB): build each, time it (tools/first_layer_split_bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
for SUB in 2 3 4; do
  touch $R/rl-rubiks_amd/csrc/rubiks_net.hip
  make -C $R/rl-rubiks_amd EXTRA="-DRC_FL_SUB=$SUB" > /tmp/build_fl_$SUB.log 2>&1 || { echo "build SUB=$SUB failed"; tail -5 /tmp/build_fl_$SUB.log; continue; }
  echo "== RC_FL_SUB=$SUB"; python3 $R/tools/first_layer_split_bench.py --quick 2>&1 | grep -v amdgpu.ids
done
