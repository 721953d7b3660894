"""
A small deterministic policy/value network used to drive search agents in parity tests.

All weights are small integers (times a power of two), inputs are one-hot and the activation is
ReLU, so every intermediate is an exactly representable integer in float32: the outputs are
identical on CPU and GPU and for every batch size / GEMM tiling.  That makes whole search trees
comparable node-for-node between the reference, the oracle and the HIP path.

It follows the call convention of the reference's Model.forward (librubiks/model.py:131-141):
net(x) -> [policy_logits, value]; net(x, policy=False) -> value; net(x, value=False) -> policy.
"""
import numpy as np
import torch
import torch.nn as nn


class StandInNet(nn.Module):
    HIDDEN = 64

    def __init__(self, seed: int = 0, weights: dict = None):
        super().__init__()
        if weights is None:
            rng = np.random.RandomState(seed)
            weights = {
                "w1": rng.randint(-2, 3, (480, self.HIDDEN)).astype(np.int8),
                "b1": rng.randint(-3, 4, (self.HIDDEN,)).astype(np.int8),
                "wp": rng.randint(-1, 2, (self.HIDDEN, 12)).astype(np.int8),
                "wv": rng.randint(-1, 2, (self.HIDDEN, 1)).astype(np.int8),
            }
            # an informative part of the value: +1 for every cubie that sits solved (code 3i / 2i),
            # so that searches actually solve shallow scrambles and exercise the solved-tree paths
            wd = np.zeros((480, 1), dtype=np.int8)
            solved = np.concatenate([3 * np.arange(8), 2 * np.arange(12)])
            wd[24 * np.arange(20) + solved, 0] = 1
            weights["wd"] = wd
        for k, v in weights.items():
            self.register_buffer(k, torch.from_numpy(np.asarray(v).astype(np.float32)))
        self.scale = 1.0 / 16.0

    def numpy_weights(self):
        return {k: getattr(self, k).cpu().numpy().astype(np.int8) for k in ("w1", "b1", "wp", "wv", "wd")}

    def forward(self, x, policy=True, value=True):
        assert policy or value
        h = torch.relu(x.float() @ self.w1 + self.b1)
        out = []
        if policy:
            out.append((h @ self.wp) * self.scale)
        if value:
            out.append((h @ self.wv) * self.scale + x.float() @ self.wd)
        return out if len(out) > 1 else out[0]
