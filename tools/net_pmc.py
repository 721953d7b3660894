"""The policy/value network forward of one MCTS step (11 264 child rows, trained weights, bf16) run eagerly a few times,
for `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model  # noqa: E402

np.random.seed(0)
model = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
cubes, _, _ = cube.scramble_batch(11264, 20, True)
eng = InferenceNet(model, torch.bfloat16)
x1 = torch.empty((11264, 4096), dtype=torch.bfloat16, device="cuda")
for _ in range(4):
    eng.head_cubes(cubes, x1)
torch.cuda.synchronize()
