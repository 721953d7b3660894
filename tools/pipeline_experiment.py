"""Experiment: K half-forests on K streams (graph replay each) vs one forest -- do latency-bound MCTS phases hide behind the other halves' GEMMs?"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.cube import DeviceCubes  # noqa: E402
from librubiks.model import InferenceNet, Model, ModelConfig  # noqa: E402
from librubiks.solving.mcts_device import MCTSForest  # noqa: E402


def run(parts, trees, model, steps=200, warm=20, c=0.6):
    np.random.seed(0)
    cubes, _, _ = cube.scramble_batch(trees, 20, True)
    per = trees // parts
    forests, streams = [], []
    for i in range(parts):
        roots = DeviceCubes.empty(per)
        roots.soa[:, :per] = cubes.soa[:, i * per:(i + 1) * per]
        f = MCTSForest(per, 12 * (steps + warm + 8) + 64)
        f.set_net(InferenceNet(model, torch.bfloat16))
        f.reset(roots)
        forests.append(f)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    for _ in range(warm):
        for f, s in zip(forests, streams):
            with torch.cuda.stream(s):
                f.step(c, f.C, use_graph=True)
    torch.cuda.synchronize()
    n0 = sum(int(f.n_nodes.sum().item()) for f in forests)
    t0 = time.perf_counter()
    for _ in range(steps):
        for f, s in zip(forests, streams):
            with torch.cuda.stream(s):
                f.step(c, f.C, use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n1 = sum(int(f.n_nodes.sum().item()) for f in forests)
    return (n1 - n0) / dt, dt / steps * 1e3


if __name__ == "__main__":
    wdir = os.path.join(ROOT, "weights", "fc_small_r1")
    for name, model in (("trained", Model.load(wdir).eval()), ("random", Model.create(ModelConfig()).eval())):
        for parts in (1, 2, 4):
            v, ms = run(parts, 1024, model)
            print(f"{name:8s} parts={parts}: {v / 1e6:.2f} M nodes/s, {ms:.4f} ms per step of all parts", flush=True)
