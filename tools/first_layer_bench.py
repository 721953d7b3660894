"""Times the fused input-layer kernel variants in isolation (12 288 rows x 4 096 columns, HIP events)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks.model import InferenceNet, Model, ModelConfig  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16", "f16pair", "bf16", "mfma", "mfma16"]
    torch.manual_seed(0)
    np.random.seed(0)
    model = Model.create(ModelConfig()).eval()
    cubes, _, _ = cube.scramble_batch(n, 20, True)
    weights = os.path.join(ROOT, "weights", "fc_small_r1")
    if os.path.isdir(weights):
        model = Model.load(weights).eval()
    ref32 = InferenceNet(model, torch.float32)
    W, b, act = ref32.layers[0]
    exact = torch.nn.functional.elu(torch.addmm(b, cubes.as_oh(torch.float32), W.t()).double())
    for v in variants:
        eng = InferenceNet(model, torch.bfloat16, first_layer_table=v)
        out = torch.empty((n, 4096), dtype=torch.bfloat16, device="cuda")
        for _ in range(3):
            eng.first_layer(cubes, out)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        ev[0].record()
        for i in range(20):
            eng.first_layer(cubes, out)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(20))
        err = (out.double() - exact).abs()
        print(f"{v}: median {ts[10] * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us | vs fp32 layer: max abs err {float(err.max()):.4f}, "
              f"mean abs err {float(err.mean()):.5f} (mean |value| {float(exact.abs().mean()):.3f})", flush=True)
