"""
CPU restatement of the reference's search agents on the hot path: BFS (BASELINE config #1), MCTS and
batch weighted A* (librubiks/solving/agents.py:92-129, 171-413, 415-645).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The agents are single-problem and sequential exactly like the reference; they sit on the NumPy
oracle of the cube environment (oracle/cube.py).  The network is a plain callable so the oracle
has no torch dependency of its own:

    net(states_int8[n,20]) -> (P float[n,12] softmaxed policy, V float[n] value)

`TorchNet` below adapts a torch module the way the reference calls it (`as_oh` -> `net(oh)` ->
`softmax(dim=1)`, agents.py:470-473,548-552,379-381).

Pinned by tests/golden/agents_golden.npz: traces (node count, action queue, neighbour table, visit
counts, values, A* G/parents) recorded from the imported reference agents driven by a small
deterministic network (tests/golden/make_golden_agents.py).

Two places define behaviour where the reference raises / hangs (both unreachable in the golden traces):
  * MCTS: a leaf whose 12 children are all known already -> reference raises ValueError on
    `v.max()` of an empty array (agents.py:559); here the best child value falls back to the max
    over the existing neighbours' V.
  * A*: empty open list -> reference would spin forever; here the search stops and returns False.
A PUCT descent is as long as it gets, exactly as in the reference (agents.py:575-595): the oracle has no path bound.  It
only RECORDS the longest descent of a search (`deepest_path`, nodes on the path) so that tests can tell which trees a
caller-chosen bound of the product's path store (MCTS(max_path=...), a resource limit, not a search parameter) would end.
"""
import heapq
from collections import deque

import numpy as np

from oracle import cube

N_ACT = cube.N_ACTIONS
_REV = np.arange(N_ACT) ^ 1


class TorchNet:
    """Wraps a torch module with the reference's call convention into the oracle's callable."""

    def __init__(self, module, device="cpu"):
        import torch
        self.torch, self.module, self.device = torch, module, device
        module.eval()

    def __call__(self, states: np.ndarray):
        torch = self.torch
        with torch.no_grad():
            oh = torch.from_numpy(cube.as_oh(states)).to(self.device)
            if len(states) == 0:
                return np.zeros((0, N_ACT), dtype=np.float32), np.zeros(0, dtype=np.float32)
            p, v = self.module(oh)
            return p.softmax(dim=1).cpu().numpy(), v.cpu().numpy().reshape(-1)

    def logits(self, states: np.ndarray):
        """Raw policy-head outputs (EGVM takes the argmax of these, agents.py:700-701)."""
        torch = self.torch
        with torch.no_grad():
            if len(states) == 0:
                return np.zeros((0, N_ACT), dtype=np.float32)
            oh = torch.from_numpy(cube.as_oh(states)).to(self.device)
            return self.module(oh, policy=True, value=False).cpu().numpy()

    def value(self, states: np.ndarray):
        torch = self.torch
        with torch.no_grad():
            if len(states) == 0:
                return np.zeros(0, dtype=np.float32)
            oh = torch.from_numpy(cube.as_oh(states)).to(self.device)
            return self.module(oh, policy=False, value=True).cpu().numpy().reshape(-1)


# =================================================================================================
# BFS  (agents.py:92-129)
# =================================================================================================
class BFS:
    def __init__(self):
        self.action_queue = deque()
        self.states = {}

    def __len__(self):
        return len(self.states)

    def search(self, state: np.ndarray, max_states: int) -> bool:
        self.action_queue = deque()
        self.states = {}
        if cube.is_solved(state):
            return True
        self.states = {state.tobytes(): (None, None)}   # key -> (parent key, action from parent)
        frontier = deque([state])
        while len(self) < max_states:
            cur = frontier.popleft()
            cur_key = cur.tobytes()
            for a, (face, d) in enumerate(cube.ACTION_SPACE):
                nxt = cube.rotate(cur, face, d)
                key = nxt.tobytes()
                if key in self.states:
                    continue
                if cube.is_solved(nxt):
                    self.action_queue.appendleft(a)
                    while self.states[cur_key][0] is not None:
                        self.action_queue.appendleft(self.states[cur_key][1])
                        cur_key = self.states[cur_key][0]
                    return True
                self.states[key] = (cur_key, a)
                frontier.append(nxt)
        return False


# =================================================================================================
# MCTS  (agents.py:415-645)
# =================================================================================================
class MCTS:
    NU = 100   # virtual loss (agents.py:433)

    def __init__(self, net, c: float, search_graph: bool, initial_capacity: int = 1000):
        self.net, self.c, self.search_graph = net, c, search_graph
        self.cap0 = initial_capacity
        self.action_queue = deque()

    def __len__(self):
        return len(self.indices)

    # ---- storage (index 0 is the "no neighbour" sentinel, agents.py:419-421,437-459) ----------
    def _reset(self):
        n = self.cap0
        self.action_queue = deque()
        self.indices = {}
        self.states = np.empty((n, 20), dtype=np.int8)
        self.neighbors = np.zeros((n, N_ACT), dtype=int)
        self.leaves = np.ones(n, dtype=bool)
        self.P = np.empty((n, N_ACT))
        self.V = np.empty(n)
        self.N = np.zeros((n, N_ACT), dtype=int)
        self.W = np.zeros((n, N_ACT))
        self.L = np.zeros((n, N_ACT))

    def _grow(self):
        n = len(self.states)
        self.states = np.concatenate([self.states, np.empty((n, 20), dtype=np.int8)])
        self.neighbors = np.concatenate([self.neighbors, np.zeros((n, N_ACT), dtype=int)])
        self.leaves = np.concatenate([self.leaves, np.ones(n, dtype=bool)])
        self.P = np.concatenate([self.P, np.empty((n, N_ACT))])
        self.V = np.concatenate([self.V, np.empty(n)])
        self.N = np.concatenate([self.N, np.zeros((n, N_ACT), dtype=int)])
        self.W = np.concatenate([self.W, np.zeros((n, N_ACT))])
        self.L = np.concatenate([self.L, np.zeros((n, N_ACT))])

    # ---- search loop (agents.py:461-494) ---------------------------------------------------------
    def search(self, state: np.ndarray, max_states: int, max_iterations: int = None) -> bool:
        self._reset()
        self.deepest_path = 1
        self.indices[state.tobytes()] = 1
        self.states[1] = state
        if cube.is_solved(state):
            return True
        p, v = self.net(state[None])
        self.P[1], self.V[1] = p[0], v[0]
        path, actions = [1], []
        self.iterations = 0
        while len(self) + N_ACT <= max_states and (max_iterations is None or self.iterations < max_iterations):
            self.iterations += 1
            solved_idx, solved_action = self._expand_leaf(path, actions)
            if solved_idx != -1:
                self.action_queue = deque(actions) + deque([solved_action])
                if self.search_graph:
                    self._complete_graph()
                    self._shorten_action_queue(solved_idx)
                return True
            path, actions = self._find_leaf()
            self.deepest_path = max(self.deepest_path, len(path))
        self.action_queue = deque(actions)   # best guess when the budget runs out (agents.py:492)
        return False

    # ---- expansion + backup (agents.py:496-573) -------------------------------------------------
    def _expand_leaf(self, path, actions):
        if len(self) + N_ACT > len(self.states):
            self._grow()
        leaf = path[-1]
        children = cube.expand12(self.states[leaf][None])
        keys = [c.tobytes() for c in children]
        unseen = np.array([k not in self.indices for k in keys])

        # unseen children get the next indices in child order (agents.py:523-529)
        new_idx = len(self) + 1 + np.arange(int(unseen.sum()))
        for k, i in zip((k for k, u in zip(keys, unseen) if u), new_idx):
            self.indices[k] = int(i)
        child_idx = np.array([self.indices[k] for k in keys])
        self.states[new_idx] = children[unseen]

        # links both ways; the leaf stops being a leaf (agents.py:533-536)
        act = np.arange(N_ACT)
        self.neighbors[leaf, act] = child_idx
        self.neighbors[child_idx, _REV] = leaf
        self.leaves[leaf] = False

        # first solved child wins (agents.py:540-543)
        solved_idx = solved_action = -1
        hit = np.flatnonzero(cube.multi_is_solved(children))
        if hit.size:
            solved_action = int(hit[0])
            solved_idx = int(child_idx[solved_action])

        # network on the NEW children only (agents.py:548-557)
        p, v = self.net(children[unseen])
        self.P[new_idx] = p
        self.V[new_idx] = v
        if len(v):
            best = v.max()
        else:
            best = self.V[self.neighbors[leaf]].max()   # defined here; the reference raises (module docstring)

        # W updates (agents.py:560-562), then N / L along the visited path (agents.py:567-570)
        self.W[leaf] = self.V[self.neighbors[leaf]]
        self.W[new_idx] = np.tile(v, (N_ACT, 1)).T
        up, down = path[:-1], path[1:]
        self.W[up, actions] = np.maximum(self.W[up, actions], best)
        if actions:
            self.N[up, actions] += 1          # buffered: a repeated (node, action) pair counts once
            self.L[up, actions] = 0
            self.L[down, _REV[np.array(actions)]] = 0
        return solved_idx, solved_action

    # ---- PUCT descent with virtual loss (agents.py:575-595) -------------------------------------
    def _find_leaf(self):
        cur = 1
        path, actions = [cur], []
        while not self.leaves[cur]:
            sqrt_n = np.sqrt(self.N[cur].sum())
            u = self.c * self.P[cur] * sqrt_n / (1 + self.N[cur])
            q = self.W[cur] - self.L[cur]
            a = int((u + q).argmax())   # first maximum
            self.L[cur, a] += self.NU
            cur = int(self.neighbors[cur, a])
            self.L[cur, a ^ 1] += self.NU
            path.append(cur)
            actions.append(a)
        return path, actions

    # ---- post-processing of a solved tree (agents.py:597-633) -----------------------------------
    def _complete_graph(self):
        leaf_idx = np.flatnonzero(self.leaves[:len(self) + 1])[1:]
        if len(leaf_idx) == 0:
            return
        children = cube.expand12(self.states[leaf_idx])
        child_idx = np.array([self.indices.get(c.tobytes(), 0) for c in children])
        rep_leaf = np.repeat(leaf_idx, N_ACT)
        act = np.tile(np.arange(N_ACT), len(leaf_idx))
        self.neighbors[rep_leaf, act] = child_idx
        self.neighbors[child_idx, _REV[act]] = rep_leaf
        self.neighbors[0] = 0

    def _shorten_action_queue(self, solved_idx: int):
        if solved_idx == 1:
            return
        self.action_queue = deque()
        came_from = {1: (None, None)}
        frontier = deque([1])
        while frontier:
            v = frontier.popleft()
            for a, n in enumerate(self.neighbors[v]):
                n = int(n)
                if not n or n in came_from:
                    continue
                if n == solved_idx:
                    self.action_queue.appendleft(a)
                    while came_from[v][0] is not None:
                        self.action_queue.appendleft(came_from[v][1])
                        v = came_from[v][0]
                    return
                came_from[n] = (v, a)
                frontier.append(n)


# =================================================================================================
# Batch weighted A*  (agents.py:171-413)
# =================================================================================================
class AStar:
    def __init__(self, net, lambda_: float, expansions: int, initial_capacity: int = 1000):
        self.net, self.lambda_, self.expansions = net, lambda_, expansions
        self.cap0 = initial_capacity
        self.action_queue = deque()

    def __len__(self):
        return len(self.indices)

    def _reset(self):
        n = self.cap0
        self.action_queue = deque()
        self.open_queue = []
        self.indices = {}
        self.states = np.empty((n, 20), dtype=np.int8)
        self.parents = np.empty(n, dtype=int)
        self.parent_actions = np.zeros(n, dtype=int)
        self.G = np.empty(n)

    def _grow(self):
        n = len(self.states)
        self.states = np.concatenate([self.states, np.empty((n, 20), dtype=np.int8)])
        self.parents = np.concatenate([self.parents, np.zeros(n, dtype=int)])
        self.parent_actions = np.concatenate([self.parent_actions, np.zeros(n, dtype=int)])
        self.G = np.concatenate([self.G, np.empty(n)])

    # agents.py:221-252
    def search(self, state: np.ndarray, max_states: int, max_iterations: int = None) -> bool:
        self._reset()
        self.open_empty = False
        if cube.is_solved(state):
            return True
        self.indices[state.tobytes()] = 1
        self.states[1] = state
        self.G[1] = 0
        heapq.heappush(self.open_queue, (0, 1))
        self.iterations = 0
        return self.resume(max_states, max_iterations)

    def resume(self, max_states: int, max_iterations: int = None) -> bool:
        """The search loop, from wherever the open list stands (tests call it again after `search(max_iterations=k)`)."""
        done = 0
        while len(self) + self.expansions * N_ACT <= max_states and (max_iterations is None or done < max_iterations):
            self.iterations += 1
            done += 1
            n_pop = min(len(self.open_queue), self.expansions)
            if n_pop == 0:
                self.open_empty = True
                return False   # defined here; the reference would spin (module docstring)
            batch = np.array([heapq.heappop(self.open_queue)[1] for _ in range(n_pop)], dtype=int)
            if self._expand_batch(batch):
                i = self.indices[cube.get_solved().tobytes()]
                while i != 1:
                    self.action_queue.appendleft(int(self.parent_actions[i]))
                    i = int(self.parents[i])
                return True
        return False

    # agents.py:254-331
    def _expand_batch(self, batch: np.ndarray) -> bool:
        while len(self) + len(batch) * N_ACT > len(self.states):
            self._grow()
        parent_of_row = np.repeat(batch, N_ACT)
        action_of_row = np.tile(np.arange(N_ACT), len(batch))
        children = cube.expand12(self.states[batch])
        keys = [c.tobytes() for c in children]

        seen = np.array([k in self.indices for k in keys])
        first = np.zeros(len(keys), dtype=bool)       # first occurrence of each distinct child in row order
        first_row = {}
        for r, k in enumerate(keys):
            if k not in first_row:
                first_row[k] = r
                first[r] = True
        first_seen, first_unseen = first & seen, first & ~seen

        new_states = children[first_unseen]
        new_idx = len(self) + 1 + np.arange(int(first_unseen.sum()))
        for k, i in zip((k for k, f in zip(keys, first_unseen) if f), new_idx):
            self.indices[k] = int(i)
        child_idx = np.array([self.indices[k] for k in keys])
        old_idx = child_idx[first_seen]
        self.states[new_idx] = new_states

        new_parent = parent_of_row[first_unseen]
        self.G[new_idx] = self.G[new_parent] + 1
        self.parent_actions[new_idx] = action_of_row[first_unseen]
        self.parents[new_idx] = new_parent
        for cost, i in zip(self.cost(new_states, new_idx), new_idx):
            heapq.heappush(self.open_queue, (cost, i))

        if cube.multi_is_solved(new_states).any():    # win check on NEW states only (agents.py:321)
            return True
        self._relax_seen(old_idx, parent_of_row[first_seen], action_of_row[first_seen])
        return False

    # agents.py:333-367
    def _relax_seen(self, state_idx, parent_idx, actions):
        better = self.G[parent_idx] + 1 < self.G[state_idx]
        s, p = state_idx[better], parent_idx[better]
        self.G[s] = self.G[p] + 1
        self.parent_actions[s] = actions[better]
        self.parents[s] = p

        shortcut = self.G[state_idx] + 1 < self.G[parent_idx]
        s, p = state_idx[shortcut], parent_idx[shortcut]
        self.G[p] = self.G[s] + 1
        self.parent_actions[p] = _REV[actions[shortcut]]
        self.parents[p] = s

    # agents.py:369-383
    def cost(self, states: np.ndarray, idx: np.ndarray) -> np.ndarray:
        h = -self.net.value(states)
        return self.lambda_ * self.G[idx] + h


# =================================================================================================
# One-step-lookahead agents and EGVM  (agents.py:14-169, 649-726)
# =================================================================================================
# The reference's Agent.search loop is `while tock() < time_limit and len(self) < max_states`, but
# len(self) is only updated when the loop ends (agents.py:30-38), so these agents are bounded by
# wall time alone.  For deterministic tests the restatement bounds the number of steps by
# `max_states` instead (same as the MI355X build); on solved games both agree with the reference.
class _StepAgent:
    def __init__(self):
        self.action_queue = deque()
        self._explored = 0

    def __len__(self):
        return self._explored

    def search(self, state: np.ndarray, max_states: int) -> bool:
        self.action_queue = deque()
        self._explored = 0
        if cube.is_solved(state):
            return True
        found = False
        while len(self.action_queue) < max_states:
            action, state, found = self._step(state)
            self.action_queue.append(action)
            if found:
                break
        self._explored = len(self.action_queue)
        return found


class PolicySearch(_StepAgent):
    """Greedy policy (agents.py:132-151, sample_policy=False)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def _step(self, state):
        p, _ = self.net(state[None])
        action = int(p[0].argmax())
        state = cube.rotate(state, *cube.ACTION_SPACE[action])
        return action, state, cube.is_solved(state)


class ValueSearch(_StepAgent):
    """Greedy value with a one-move solution check (agents.py:154-169)."""

    def __init__(self, net):
        super().__init__()
        self.net = net

    def _step(self, state):
        children = cube.expand12(state[None])
        hit = np.flatnonzero(cube.multi_is_solved(children))
        if hit.size:
            return int(hit[0]), children[hit[0]], True
        action = int(np.argmax(self.net.value(children)))
        return action, children[action], False


class EGVM:
    """Epsilon-greedy value maximisation (agents.py:649-726); draws from the global np.random stream."""

    def __init__(self, net, epsilon: float, workers: int, depth: int):
        self.net, self.epsilon, self.workers, self.depth = net, epsilon, workers, depth
        self.action_queue = deque()
        self._explored = 0

    def __len__(self):
        return self._explored

    def search(self, state: np.ndarray, max_states: int) -> bool:
        self.action_queue = deque()
        self._explored = 0
        if cube.is_solved(state):
            return True
        while len(self) + self.workers * self.depth <= max_states:
            paths, states, solved = self._expand(state)
            if solved is not None:
                w, d = solved
                self.action_queue += deque(int(a) for a in paths[w, :d])
                return True
            best = int(np.argmax(self.net.value(states)))   # rows are worker-major: row w * depth + d (agents.py:681-682,714)
            state = states[best]
            worker, depth = best // self.depth, best % self.depth
            self.action_queue += deque(int(a) for a in paths[worker, :depth + 1])
        return False

    def _expand(self, state):
        states = cube.repeat_state(state, self.workers)
        paths = np.empty((self.workers, self.depth), dtype=int)
        visited = np.empty((self.workers * self.depth, 20), dtype=np.int8)
        for d in range(self.depth):
            use_random = np.random.choice(2, self.workers, p=[1 - self.epsilon, self.epsilon]).astype(bool)
            actions = np.empty(self.workers, dtype=int)
            actions[use_random] = np.random.randint(0, N_ACT, use_random.sum())
            if (~use_random).any():
                actions[~use_random] = self.net.logits(states[~use_random]).argmax(axis=1)
            paths[:, d] = actions
            states = cube.multi_rotate_actions(states, actions)
            hit = np.flatnonzero(cube.multi_is_solved(states))
            if hit.size:
                self._explored += (d + 1) * self.workers
                return paths, None, (int(hit[0]), d + 1)
            visited[np.arange(self.workers) * self.depth + d] = states   # agents.py:681-682,714
        self._explored += len(visited)
        return paths, visited, None
