"""Micro-benchmark of the environment kernels on one MI355X: achieved algorithmic GB/s vs the HBM roof."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks.cube import DeviceCubes  # noqa: E402

HBM_PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] * 1e-3, ts[0] * 1e-3


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = 1 << logn
    g = torch.Generator(device="cuda").manual_seed(0)
    cubes = DeviceCubes.solved(n)
    for _ in range(30):
        cubes = cubes.multi_rotate(torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g))
    act = torch.randint(0, 12, (n,), dtype=torch.uint8, device="cuda", generator=g)
    out = DeviceCubes.empty(n)
    res = {}

    def report(name, unit_bytes, units, t):
        med, best = t
        res[name] = {"units": units, "ms_median": med * 1e3, "GBps": unit_bytes * units / med / 1e9,
                     "frac_of_8TBps": unit_bytes * units / med / 1e9 / HBM_PEAK, "Munits_per_s": units / med / 1e6}

    report("multi_rotate", 41, n, timed(lambda: cubes.multi_rotate(act, out=out)))
    np_ = n // 4
    parents = DeviceCubes(cubes.soa[:, :max(256, np_)].contiguous(), np_)
    kids = DeviceCubes.empty(12 * np_)
    report("expand12", 260, np_, timed(lambda: parents.expand12(out=kids)))
    report("is_solved_flags", 21, n, timed(lambda: cubes.is_solved()))
    report("is_solved_mask", 20.125, n, timed(lambda: cubes.solved_mask()))
    no = n // 16
    small = DeviceCubes(cubes.soa[:, :max(256, no)].contiguous(), no)
    oh32 = torch.empty((no, 480), dtype=torch.float32, device="cuda")
    oh16 = torch.empty((no, 480), dtype=torch.bfloat16, device="cuda")
    report("as_oh_f32", 1940, no, timed(lambda: small.as_oh(out=oh32)))
    report("as_oh_bf16", 980, no, timed(lambda: small.as_oh(out=oh16)))
    # a plain device copy of the same bytes as the practical roof
    a = torch.empty(41 * n // 2, dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    med, _ = timed(lambda: b.copy_(a))
    res["torch_copy_same_bytes"] = {"GBps": 2 * a.numel() / med / 1e9}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
