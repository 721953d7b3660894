"""
The allocation pattern of round 4's two GPU memory access faults, on today's node store, with the event log on:
results-only forests (the copies of finished trees that are turned into results on the side stream) as RESERVED ADDRESS RANGES,
reserved / mapped for every harvest and RELEASED (unmapped, address range back to its size class) as soon as their results have
been read -- next to the running search of one GPU's share of BASELINE configs[4] (8 192 depth-24 trees, max_states 175 000,
continuous batching from a pool).  Round 4 freed the addresses without flushing the GPU's translations; today's rc_vmm_release
flushes, and the next reservation of the class maps memory at the same addresses.  The run must (1) not fault, (2) return exactly
the games of the production configuration (results-only forests as ordinary allocations), and (3) leave a log from which
tools/vmm_classify.py can say what any address was at any time.

    RUBIKS_VMM_LOG=gpurun_out/vmm_churn.log python tools/vmm_churn_probe.py [pool factor] [trees]
"""
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import cube  # noqa: E402
from librubiks._vmm import VmmArray  # noqa: E402
from librubiks.model import Model  # noqa: E402
from librubiks.solving import mcts_device as md  # noqa: E402
from librubiks.solving.agents import MCTS  # noqa: E402

factor = int(sys.argv[1]) if len(sys.argv) > 1 else 2
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
CAP = 175000
net = Model.load(os.path.join(ROOT, "weights", "fc_small_r1")).eval()
np.random.seed(0)
pool, _, _ = cube.scramble_batch(factor * trees, 24, True)


def run(churn: bool):
    stats = collections.Counter()
    init, close = md.MCTSForest.__init__, md.MCTSForest.close
    if churn:
        def init_churn(self, n_trees, capacity, max_path=4096, device=None, _results_only=False, vmm=None):
            if _results_only:
                vmm = True                      # round 4's layout: the copies live in reserved ranges of their own
                stats["results_forests"] += 1
            init(self, n_trees, capacity, max_path, device, _results_only, vmm)

        def close_churn(self):
            if getattr(self, "results_only", False) and getattr(self, "vmm", False) and getattr(self, "_ranges", None) is not None:
                torch.cuda.synchronize()
                arrays = [a for a, _ in self._ranges.values()] + ([self._ranges_bfs] if getattr(self, "_ranges_bfs", None) is not None else [])
                for name in list(self._ranges) + list(md._NODE_FIELDS) + ["bfs"]:      # every tensor that views the ranges
                    if hasattr(self, name):
                        delattr(self, name)
                self._graphs, self._graph_pool = {}, None
                self._ranges, self._ranges_bfs = None, None
                for a in arrays:                 # released at once, as round 4 did: unmap, (today) flush, address range to its class
                    stats["released_bytes"] += a.mapped_bytes
                    a.close()
                    stats["releases"] += 1
                return
            close(self)
        md.MCTSForest.__init__, md.MCTSForest.close = init_churn, close_churn
    try:
        agent = MCTS(net, c=0.6, search_graph=True, net_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = agent.search_batch(pool, None, CAP, slots=trees)
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        if agent.forest is not None:
            agent.forest.close()
        del agent
    finally:
        md.MCTSForest.__init__, md.MCTSForest.close = init, close
    torch.cuda.synchronize()
    VmmArray.trim()
    torch.cuda.empty_cache()
    return res, secs, stats


plain, t_plain, _ = run(False)
print(f"production layout: {len(pool.numpy())} games on {trees} slots in {t_plain:.1f} s, solved {plain.solved.mean():.4f}, nodes {int(plain.nodes.sum())}", flush=True)
churn, t_churn, st = run(True)
print(f"results forests in reserved ranges, released per harvest: {t_churn:.1f} s, solved {churn.solved.mean():.4f}, nodes {int(churn.nodes.sum())}; "
      f"{st['results_forests']} results forests, {st['releases']} ranges released with {st['released_bytes'] / 1e9:.1f} GB mapped in all", flush=True)
same = (np.array_equal(plain.solved, churn.solved) and np.array_equal(plain.nodes, churn.nodes) and np.array_equal(plain.lengths, churn.lengths)
        and all(list(a) == list(b) for a, b in zip(plain.queues, churn.queues)))
print("games identical to the production layout's:", same, "; idle address space now", VmmArray.retired_bytes() >> 20, "MiB", flush=True)
sys.exit(0 if same else 1)
