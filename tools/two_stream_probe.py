"""Do two half-size forests on two HIP streams overlap one forest's tree kernels with the other's GEMMs?"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=8192)
    ap.add_argument("--slots", type=int, default=1024)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    from librubiks import cube
    from librubiks.cube.device import DeviceCubes
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(args.games, 20, True)
    model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
    dt = {"bf16": torch.bfloat16, "f32s": F32_SPLIT, "f32": torch.float32}[args.dtype]
    for groups in (1, args.groups):
        G, S = args.games // groups, args.slots // groups
        agents = [MCTS(model, c=0.6, search_graph=True, net_dtype=dt) for _ in range(groups)]
        streams = [torch.cuda.Stream() for _ in range(groups)]
        parts = []
        for g in range(groups):
            part = DeviceCubes.empty(G)
            part.soa[:, :G] = cubes.soa[:, g * G:(g + 1) * G]
            parts.append(part)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runs = []
        for a, s, p in zip(agents, streams, parts):
            with torch.cuda.stream(s):
                runs.append(a.start_batch(p, None, 50000, slots=S))
        while not all(r.done for r in runs):
            for r, s in zip(runs, streams):
                if not r.done:
                    with torch.cuda.stream(s):
                        r.round()
        nodes = 0
        for r, s in zip(runs, streams):
            with torch.cuda.stream(s):
                res = r.finish()
            nodes += int(res.nodes.sum())
        torch.cuda.synchronize()
        dtm = time.perf_counter() - t0
        print(f"{args.dtype} groups={groups} slots/group={S}: {nodes} nodes in {dtm:.3f} s = {nodes / dtm / 1e6:.2f} M/s, "
              f"iterations {[r.it for r in runs]}", flush=True)
        del agents, runs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
