"""
Shared by the pieces of bench.py (tools/benchlib/*): constants of the roofline (MI355X_MICROARCH.md), the stored PMC figures the line
cites, progress lines and HIP-event timing.  bench.py at the repository root is the entry point and owns the contract (metric, timed
region, the JSON line) and the cpu_baseline leg -- the only place besides tests/ and smoke() that imports oracle/.
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, "rl-rubiks_amd")) if p not in sys.path]

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16
MFMA_F32_PEAK_TFLOPS = 157.3
# `traffic` of roofline_env is NOT measured inside this run: it is the stored figure of separate rocprofv3 --pmc passes
# (FETCH_SIZE / WRITE_SIZE, gfx950 corrections applied by tools/rocprof_summary.py traffic) of the same launches at 2^24 states
_PROFILES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
# the most recent measurement of the kernels as they are in this tree (re-taken every round the kernels change: tools/r6_pmc_pass.sh)
PMC_FILE = next((f for f in ("r6_env_pmc_traffic.json", "r4k_env_pmc_traffic.json") if os.path.exists(os.path.join(_PROFILES, f))), "r6_env_pmc_traffic.json")
GEMM_PMC_FILE = next((f for f in ("r6_split_gemm_traffic.json", "r4_split_gemm_traffic.json") if os.path.exists(os.path.join(_PROFILES, f))),
                     "r6_split_gemm_traffic.json")   # the hidden-layer kernel that runs today (k_split_gemm<2, 4, 11, 4, 2, 0>)
PMC_SOURCE = f"stored PMC figure: profiles/{PMC_FILE} (separate rocprofv3 --pmc passes of these launches, not this run)"


_T0 = time.perf_counter()


def progress(what):
    """One short line per finished leg on stderr: where a run was when something went wrong, and a sign of life for the box's
    silence watchdog.  (The final stdout line stays the last thing printed.)"""
    free, total = torch.cuda.mem_get_info()
    print(f"[bench {time.perf_counter() - _T0:6.1f} s] {what}; HBM in use {(total - free) / 1e9:.0f} GB", file=sys.stderr, flush=True)


def event_ms(fn, reps, warm=2):
    for _ in range(warm):
        fn()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        fn()
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    return float(np.mean(ts)), float(np.min(ts))


LEG_DTYPE = {   # leg name -> the arithmetic the network computes in (`dtype` of the JSON line)
    "f32s": "f32 (f16x3 split: three f16 MFMA products per layer, fp32 accumulate)",
    "f32": "f32",
    "bf16": "bf16",
}
SPREAD_WINDOWS = 5
