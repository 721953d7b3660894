"""
Device-resident ADI (librubiks.train.Train) against the reference's recorded targets and the oracle,
plus a short end-to-end training run.  Integer outputs exact; value targets exact as well because the
stand-in net's outputs are exact in fp32.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

from oracle import agents as oa  # noqa: E402  (checker only)
from oracle import cube as oc  # noqa: E402
from oracle import train as ot  # noqa: E402


def _trainer(games, depth, method, **kw):
    from librubiks.train import Train
    args = dict(rollouts=1, batch_size=10, rollout_games=games, rollout_depth=depth, optim_fn=torch.optim.Adam,
                alpha_update=0, lr=1e-3, gamma=1, update_interval=0, agent=None, evaluator=None, evaluation_interval=0,
                with_analysis=False, tau=1, reward_method=method)
    args.update(kw)
    return Train(**args)


@pytest.mark.parametrize("method", ["paper", "lapanfix", "schultzfix", "reward0"])
def test_adi_targets_golden(method, standin_net):
    g = np.load(f"{GOLDEN}/adi_golden.npz")
    net = standin_net.cuda()
    for games, depth in ((6, 10), (5, 7)):
        pre = f"adi_{method}_{games}x{depth}_"
        np.random.seed(31)
        oh, pol, val, w = _trainer(games, depth, method).ADI_traindata(net, float(g[pre + "alpha"][0]))
        assert oh.is_cuda and oh.dtype == torch.float32 and oh.shape == (games * depth, 480)
        assert np.array_equal(np.nonzero(oh.cpu().numpy())[1].reshape(-1, 20), g[pre + "ohcols"])
        assert pol.dtype == torch.int64 and np.array_equal(pol.cpu().numpy(), g[pre + "policy"])
        assert np.array_equal(val.cpu().numpy(), g[pre + "value"])
        assert np.array_equal(w.cpu().numpy(), g[pre + "weights"])


@pytest.mark.parametrize("method", ["paper", "lapanfix", "schultzfix", "reward0"])
def test_adi_targets_vs_oracle_large(method, standin_net):
    """BASELINE config #4 shape per GPU: 2 048 states x 12 substates, chunked value calls."""
    net = standin_net.cuda()
    tr = _trainer(64, 32, method)
    tr.adi_chunk = 4096   # force several chunks
    np.random.seed(5)
    oh, pol, val, w = tr.ADI_traindata(net, 0.3)
    np.random.seed(5)
    states, rpol, rval, rw = ot.adi_traindata(oa.TorchNet(net, device="cuda").value, 64, 32, method, 0.3)
    assert np.array_equal(np.nonzero(oh.cpu().numpy())[1].reshape(-1, 20), oc.oh_indices(states))
    assert np.array_equal(pol.cpu().numpy(), rpol)
    assert np.array_equal(val.cpu().numpy(), rval)
    assert np.allclose(w.cpu().numpy(), rw, rtol=1e-6)


def test_adi_bf16_engine_close_to_fp32():
    """Value targets from the bf16 inference engine stay within 5e-2 of the fp32 module's (values are O(1))."""
    from librubiks.model import Model, ModelConfig
    torch.manual_seed(0)
    net = Model.create(ModelConfig()).eval()
    np.random.seed(1)
    _, p32, v32, _ = _trainer(32, 20, "lapanfix").ADI_traindata(net, 0.0)
    np.random.seed(1)
    _, p16, v16, _ = _trainer(32, 20, "lapanfix", adi_net_dtype=torch.bfloat16).ADI_traindata(net, 0.0)
    assert torch.allclose(v32, v16, atol=5e-2)
    assert float((p32 == p16).float().mean()) > 0.9   # near-ties may flip the argmax


def test_short_training_run():
    """Three rollouts of the full loop (ADI -> minibatch SGD -> evaluation) on fc_small."""
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    from librubiks.solving.evaluation import Evaluator
    from librubiks.train import Train
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig())
    before = net.get_params().clone()
    agent = MCTS(net, c=0.6, search_graph=False)
    ev = Evaluator(n_games=16, scrambling_depths=range(1, 3), max_time=None, max_states=200)
    tr = Train(rollouts=3, batch_size=256, rollout_games=64, rollout_depth=12, optim_fn=torch.optim.Adam, alpha_update=0.5,
               lr=1e-4, gamma=0.5, update_interval=1, agent=agent, evaluator=ev, evaluation_interval=2,
               tau=0.3, reward_method="lapanfix")
    net, best = tr.train(net)
    assert np.isfinite(tr.train_losses).all() and (tr.train_losses > 0).all()
    assert not torch.equal(before, net.get_params())
    # schedule of the reference (train.py:63-73): arange(0, 3, 2) - 1 -> [0, 1], plus the last rollout
    assert list(tr.evaluation_rollouts) == [0, 1, 2] and len(tr.sol_percents) == 3
    assert all(0 <= s <= 1 for s in tr.sol_percents)
    assert isinstance(best, Model)


def _train_rank(rank, world, port, q):
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here), os.path.join(os.path.dirname(here), "rl-rubiks_amd")]
    from ranks import init_gloo
    init_gloo(rank, world, port, seconds=120)
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    from librubiks.train import Train
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig())
    tr = Train(rollouts=2, batch_size=128, rollout_games=32, rollout_depth=8, optim_fn=torch.optim.Adam, alpha_update=0,
               lr=1e-4, gamma=1, update_interval=0, agent=MCTS(net, c=0.6, search_graph=False), evaluator=None,
               evaluation_interval=0, tau=1, reward_method="lapanfix")
    net, _ = tr.train(net)
    first_draw = int(np.random.randint(0, 2 ** 31))      # the global NumPy stream after training
    q.put((rank, tr.rollout_games, net.get_params().double().sum().item(), net.get_params()[:64].cpu().tolist(), first_draw,
           tr.train_losses.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_two_ranks():
    """Config #4 layout: ranks generate different games (half each), average gradients, keep identical weights."""
    from ranks import run_ranks
    got = run_ranks(_train_rank, 2, lambda r, port, q: (r, 2, port, q), timeout=300)
    (_, g0, sum0, head0, draw0, loss0), (_, g1, sum1, head1, draw1, loss1) = got
    assert g0 == g1 == 16                                  # 32 games per rollout split over two ranks
    assert sum0 == sum1 and head0 == head1                 # the same averaged gradients -> the same weights
    assert loss0 != loss1                                  # ... from different games (rank-private ADI streams)
    assert draw0 == draw1                                  # the global stream stays common: evaluation scrambles are shared


def _train_eval_rank(rank, world, port, q, out_dir):
    import os
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [here, os.path.dirname(here), os.path.join(os.path.dirname(here), "rl-rubiks_amd")]
    from ranks import init_gloo
    init_gloo(rank, world, port, seconds=120)
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import MCTS
    from librubiks.solving.evaluation import Evaluator
    from librubiks.train import Train
    torch.manual_seed(0)
    np.random.seed(0)
    net = Model.create(ModelConfig())
    evaluator = Evaluator(10, [2, 4], None, 150)
    seen = []
    plain_eval = evaluator.eval

    def recording_eval(agent):      # what the common np.random stream was before the evaluation, and what came out
        state = np.random.get_state()
        out = plain_eval(agent)
        seen.append((state, out[0].copy(), out[1].copy()))
        return out

    evaluator.eval = recording_eval
    tr = Train(rollouts=2, batch_size=24, rollout_games=16, rollout_depth=6, optim_fn=torch.optim.Adam, alpha_update=0,
               lr=1e-4, gamma=1, update_interval=0, agent=MCTS(net, c=0.6, search_graph=True, net_dtype=torch.float32),
               evaluator=evaluator, evaluation_interval=1, tau=1, reward_method="lapanfix")
    net, _ = tr.train(net)
    wait = tr.tt.profiles["Gradient all-reduce wait"]
    loop = tr.tt.profiles["Training loop"]
    if rank == 0:
        net.save(out_dir)
        np.save(os.path.join(out_dir, "rng_keys.npy"), np.stack([s[0][1] for s in seen]))
        np.save(os.path.join(out_dir, "rng_pos.npy"), np.array([s[0][2] for s in seen]))
    q.put((rank, [s[1].tolist() for s in seen], [s[2].tolist() for s in seen], tr.sol_percents, len(wait), wait.sum() / loop.sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_evaluates_one_common_scramble_set(tmp_path):
    """
    Two ranks train data-parallel (private ADI streams, bucketed gradient averaging) and evaluate inside the loop:
    the Evaluator shards ONE scramble set over the ranks, so both ranks report the same (depths x games) results, and
    those are what a single process gets for the final weights from the same position of the common np.random stream.
    """
    from librubiks.model import Model
    from librubiks.solving.agents import MCTS
    from librubiks.solving.evaluation import Evaluator
    from ranks import run_ranks
    got = run_ranks(_train_eval_rank, 2, lambda r, port, q: (r, 2, port, q, str(tmp_path)), timeout=300)
    (_, res0, st0, sol0, n_wait0, share0), (_, res1, st1, sol1, n_wait1, share1) = got
    assert res0 == res1 and st0 == st1 and sol0 == sol1 and len(res0) == 2          # two evaluations, identical on both ranks
    assert n_wait0 == n_wait1 == 4                                                   # 2 rollouts x 2 optimizer steps
    print(f"gradient all-reduce wait = {share0:.1%} / {share1:.1%} of the training loop (two gloo ranks on one GPU)")
    assert 0 < share0 < 1
    # single process, final weights, the stream position of the LAST evaluation
    keys, pos = np.load(tmp_path / "rng_keys.npy"), np.load(tmp_path / "rng_pos.npy")
    np.random.set_state(("MT19937", keys[-1], int(pos[-1]), 0, 0.0))
    net = Model.load(str(tmp_path)).eval()
    res, states, _ = Evaluator(10, [2, 4], None, 150).eval(MCTS(net, c=0.6, search_graph=True, net_dtype=torch.float32))
    assert res.tolist() == res0[-1] and states.tolist() == st0[-1]


def test_reference_train_test_restated():
    """tests/test_train.py:13-24 of the reference (its arguments, minus the analysis / plot that are out of scope)."""
    from librubiks.model import Model, ModelConfig
    from librubiks.solving.agents import PolicySearch
    from librubiks.solving.evaluation import Evaluator
    from librubiks.train import Train
    torch.manual_seed(42)
    np.random.seed(42)
    net = Model.create(ModelConfig())
    evaluator = Evaluator(2, max_time=.02, max_states=None, scrambling_depths=[2])
    train = Train(rollouts=2, batch_size=2, tau=0.1, alpha_update=.5, gamma=1, rollout_games=2, rollout_depth=3,
                  optim_fn=torch.optim.Adam, agent=PolicySearch(None), lr=1e-6, evaluation_interval=1, evaluator=evaluator,
                  update_interval=1, with_analysis=False, reward_method="schultzfix")
    net, min_net = train.train(net)
    assert isinstance(net, Model) and isinstance(min_net, Model)
    assert len(train.train_losses) == 2 and np.isfinite(train.train_losses).all()
    assert len(train.sol_percents) == len(train.evaluation_rollouts) and all(0 <= s <= 1 for s in train.sol_percents)
