"""
Small host utilities with the reference's names (librubiks/utils/__init__.py:14-30,
librubiks/utils/ticktock.py:53-146, librubiks/utils/logger.py:65-77): seeding, the Bernoulli
confidence half-width used for solve rates, a section profiler and a null logger.
"""
import random
from time import perf_counter

import numpy as np
import torch


def set_seeds():
    """All RNGs to 0 (reference utils/__init__.py:14-20)."""
    torch.manual_seed(0)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(0)
    np.random.seed(0)
    random.seed(0)


_Z = {0.1: 1.6448536269514722, 0.05: 1.959963984540054, 0.01: 2.5758293035489004}


def bernoulli_error(p: float, n: int, alpha: float, stringify: bool = False):
    """z * sqrt(p(1-p)/n) (reference utils/__init__.py:24-30)."""
    if alpha in _Z:
        z = _Z[alpha]
    else:
        from scipy.stats import norm
        z = norm.ppf(1 - alpha / 2)
    err = z * np.sqrt(p * (1 - p) / n)
    return f"+/- {err * 100:.0f} %" if stringify else err


class NullLogger:
    def __call__(self, *args, **kwargs):
        pass

    log = section = verbose = is_verbose = __call__


class _Section:
    def __init__(self, name, depth):
        self.name, self.depth, self.hits, self.start = name, depth, [], 0.0

    def sum(self):
        return sum(self.hits)

    def mean(self):
        return self.sum() / len(self.hits) if self.hits else 0.0

    def __len__(self):
        return len(self.hits)


class TickTock:
    """
    Nested wall-clock section timer (reference utils/ticktock.py:53-146).  With `sync=True`
    sections are bracketed by torch.cuda.synchronize() so that they measure GPU work, which the
    reference's timer cannot.
    """

    def __init__(self, sync: bool = False):
        self.sync = sync
        self._start = 0.0
        self.reset()

    def _now(self):
        if self.sync and torch.cuda.is_available():
            torch.cuda.synchronize()
        return perf_counter()

    def tick(self):
        self._start = self._now()
        return self._start

    def tock(self):
        return self._now() - self._start

    def profile(self, name: str):
        if name not in self.profiles:
            self.profiles[name] = _Section(name, self._depth)
        self._depth += 1
        self._latest = name
        self.profiles[name].start = self._now()

    def end_profile(self, name: str = None):
        end = self._now()
        name = name or self._latest
        dt = end - self.profiles[name].start
        self.profiles[name].hits.append(dt)
        self._depth -= 1
        return dt

    def reset(self):
        self.profiles = {}
        self._depth = 0
        self._latest = None

    def __str__(self):
        rows = [("Execution times", "Total time", "Hits", "Avg. time")]
        for name, s in self.profiles.items():
            rows.append(("- " * s.depth + name, f"{s.sum():.3f} s", f"{len(s):,}", f"{s.mean() * 1e3:.3f} ms"))
        widths = [max(len(r[i]) for r in rows) for i in range(4)]
        return "\n".join(" | ".join(c.ljust(w) if i == 0 else c.rjust(w) for i, (c, w) in enumerate(zip(r, widths)))
                         for r in rows)
