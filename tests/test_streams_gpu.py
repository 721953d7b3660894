"""
The product launches on torch's CURRENT stream (librubiks/_hip.py: stream_ptr) and keeps its own side stream for results
(MCTSRun).  A caller that overlaps a search with other work runs it inside `with torch.cuda.stream(s)`: everything -- the
kernels of the C ABI, the captured HIP graphs, the node store's growth, the copies of results -- must follow, and the results
must be those of the default stream.  Here every batched agent is run on the default stream and then on a side stream, while
the default stream is kept busy with unrelated work that a missing dependency would race against.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

from oracle import cube as oc  # noqa: E402

WEIGHTS = os.path.join(ROOT, "weights", "fc_small_r1")


def _same(a, b):
    assert np.array_equal(a.solved, b.solved) and np.array_equal(a.lengths, b.lengths) and np.array_equal(a.nodes, b.nodes)
    assert all(list(x) == list(y) for x, y in zip(a.queues, b.queues))


def _busy(x):
    """Unrelated work queued on the default stream: ~10 ms of GEMMs."""
    for _ in range(20):
        x = x @ x
        x = x / x.abs().max()
    return x


@pytest.mark.parametrize("agent_kind", ["mcts_refill", "mcts_batch", "astar"])
def test_agents_on_a_side_stream_give_the_default_stream_results(agent_kind):
    from librubiks.model import F32_SPLIT, Model
    from librubiks.solving.agents import MCTS, AStar
    if not os.path.isdir(WEIGHTS):
        pytest.skip("needs the trained weights")
    net = Model.load(WEIGHTS).eval()
    np.random.seed(77)
    states = np.array([oc.scramble(11 + i % 7, True)[0] for i in range(160)])

    def run():
        if agent_kind == "astar":
            return AStar(net, lambda_=0.2, expansions=64, net_dtype=F32_SPLIT).search_batch(states[:96], None, 6000)
        agent = MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT, sync_every=8)
        if agent_kind == "mcts_refill":
            return agent.search_batch(states, None, 4000, slots=64)
        return agent.search_batch(states[:128], None, 4000)

    ref = run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    x = torch.randn(2048, 2048, device="cuda")
    y = _busy(x)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got = run()
    y = _busy(y)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    _same(ref, got)
    assert 0.3 < ref.solved.mean()          # a meaningful search: most MCTS games and about half of the A* ones are solved


def _mcts_inputs():
    from librubiks.model import Model
    if not os.path.isdir(WEIGHTS):
        pytest.skip("needs the trained weights")
    np.random.seed(78)
    a = np.array([oc.scramble(10 + i % 8, True)[0] for i in range(96)])
    b = np.array([oc.scramble(12 + i % 5, True)[0] for i in range(72)])
    return Model.load(WEIGHTS).eval(), a, b


def test_two_searches_interleaved_round_by_round_do_not_disturb_each_other():
    """Two agents, two forests, their rounds alternating on one stream (an evaluation that keeps two searches going, a service
    with two requests): nothing in the library is shared between forests but read-only tables, so each must end as it ends alone."""
    from librubiks.model import F32_SPLIT
    from librubiks.solving.agents import MCTS
    net, sa, sb = _mcts_inputs()
    mk = lambda: MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT, sync_every=8)   # noqa: E731
    alone_a = mk().search_batch(sa, None, 4000, slots=48)
    alone_b = mk().search_batch(sb, None, 3000)
    ra, rb = mk().start_batch(sa, None, 4000, slots=48), mk().start_batch(sb, None, 3000)
    while not (ra.done and rb.done):
        if not ra.done:
            ra.round()
        if not rb.done:
            rb.round()
    _same(alone_a, ra.finish())
    _same(alone_b, rb.finish())


def test_two_host_threads_with_their_own_streams_search_at_the_same_time():
    """One host thread per search, each on its own stream, agents prepared (forest, engine, HIP graphs) one after the other
    beforehand -- capturing a graph is the one step that does not tolerate another thread's launches.  ctypes releases the GIL
    around every C-ABI call, so the two threads really are inside the library together."""
    import threading
    from librubiks.model import F32_SPLIT
    from librubiks.solving.agents import MCTS
    net, sa, sb = _mcts_inputs()
    mk = lambda: MCTS(net, c=0.6, search_graph=True, net_dtype=F32_SPLIT, sync_every=8)   # noqa: E731
    alone = [mk().search_batch(sa, None, 4000), mk().search_batch(sb, None, 3000)]
    agents, jobs = [mk(), mk()], [(sa, 4000), (sb, 3000)]
    for agent, (s, cap) in zip(agents, jobs):
        agent.prepare(len(s), cap)
    torch.cuda.synchronize()
    out, errors = [None, None], []

    def work(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                out[i] = agents[i].search_batch(jobs[i][0], None, jobs[i][1])
            stream.synchronize()
        except BaseException as e:   # noqa: BLE001 -- reported by the test
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors and not any(t.is_alive() for t in threads), errors
    _same(alone[0], out[0])
    _same(alone[1], out[1])
