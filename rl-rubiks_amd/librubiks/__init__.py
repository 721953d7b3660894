"""
librubiks -- MI355X-native drop-in for the hot path of peleiden/rl-rubiks.

Same import surface as the reference package for that path (`from librubiks import cube, gpu`;
`from librubiks.solving.agents import MCTS, AStar`; `from librubiks.model import Model, ModelConfig`),
with the cube environment and the search loops running as hand-written HIP kernels for gfx950
behind the C ABI declared in include/rubiks_hip.h.

Mirrors reference librubiks/__init__.py:5-21 (`cpu`, `gpu`, `reset_cuda`, `no_grad`).
There is NO CPU implementation of the cube environment in this package: without an MI355X and the
built librubiks_hip.so every cube / agent call raises.
"""
import torch

_HAS_GPU = torch.cuda.is_available()
cpu, gpu = torch.device("cpu"), torch.device("cuda" if _HAS_GPU else "cpu")


def reset_cuda() -> None:
    """Returns cached allocator blocks to the driver and waits for the device (reference: same name, same effect)."""
    torch.cuda.empty_cache()
    if _HAS_GPU:
        torch.cuda.synchronize()


def no_grad(fun):
    """Decorator: run `fun` without autograd (torch.no_grad used as a decorator keeps the wrapped signature)."""
    return torch.no_grad()(fun)
