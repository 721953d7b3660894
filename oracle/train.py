"""
NumPy restatement of the reference's ADI target generation, Train.ADI_traindata
(librubiks/train.py:257-339).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Pinned by tests/golden/adi_golden.npz (recorded from the imported reference, all four reward methods).
"""
import numpy as np

from oracle import cube


def adi_traindata(value_fn, games: int, depth: int, reward_method: str, alpha: float):
    """
    value_fn(states int8[n,20]) -> float32[n] value-head outputs of the generator network (eval mode).
    Returns (states int8[G*D,20], policy_targets int64[G*D], value_targets float32[G*D], loss_weights float32[G*D]).
    """
    # scrambled sequences; the solved cube is part of them only with Lapan's fix (train.py:277)
    states, _ = cube.sequence_scrambler(games, depth, with_solved=reward_method == "lapanfix")
    solved_states = cube.multi_is_solved(states)
    substates = cube.expand12(states)                                   # train.py:285
    solved_sub = cube.multi_is_solved(substates)
    win = np.float32(0 if reward_method == "reward0" else 1)            # train.py:294-296
    rewards = np.where(solved_sub, win, np.float32(-1)).astype(np.float32)
    values = (value_fn(substates).astype(np.float32) + rewards).reshape(-1, 12)   # train.py:303,314-315
    policy = values.argmax(axis=1)                                      # first maximum
    value_targets = values[np.arange(len(values)), policy].copy()
    if reward_method == "lapanfix":
        value_targets[solved_states] = 0                                # train.py:318-320
    elif reward_method == "schultzfix":
        value_targets[np.arange(0, len(states), depth)] = 0             # train.py:321-325
    weighted = np.tile(1 / np.arange(1, depth + 1), games)              # train.py:330-333
    ws, us = weighted.sum(), len(weighted)
    loss_weights = ((1 - alpha) * weighted / ws + alpha * np.ones_like(weighted) / us) * (ws + us)
    return states, policy.astype(np.int64), value_targets, loss_weights.astype(np.float32)
