"""
Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (peleiden/rl-rubiks).

Run only in the build container, where the reference is mounted read-only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        python /root/repo/tests/golden/make_golden.py [cube] [bfs] [bfs_cut] [agents] [adi] [simple] [model]

The fixtures are DATA (inputs + the reference's outputs).  No reference source travels with them.
The GPU box never runs this script (it has no /root/reference); it only reads the committed files.
"""
import json
import os
import sys
import warnings

import numpy as np

warnings.filterwarnings("ignore")
OUT = os.path.dirname(os.path.abspath(__file__))


def _reachable_states(cube, n, moves, seed):
    """n states, each `moves` random moves away from solved (reference multi_rotate does the moving)."""
    rng = np.random.RandomState(seed)
    states = np.tile(cube.get_solved(), (n, 1))
    for _ in range(moves):
        states = cube.multi_rotate(states, rng.randint(0, 6, n), rng.randint(0, 2, n))
    return states


def make_cube():
    from librubiks import cube
    from librubiks.cube.cube import _Cube2024
    assert cube.get_is2024()
    fx = {}

    # (1) move tables, (2) solved state
    fx["maps"] = _Cube2024.maps.copy()
    assert fx["maps"].shape == (2, 6, 2, 24) and fx["maps"].dtype == np.int8
    with open("/root/reference/frontend/src/assets/maps.json") as f:
        mj = json.load(f)
    assert np.array_equal(np.array(mj["map_neg"]), fx["maps"][0])
    assert np.array_equal(np.array(mj["map_pos"]), fx["maps"][1])
    fx["solved"] = cube.get_solved()

    # (3) scrambles: seeds {0, 42}, depths {20, 24}, first 16 games in call order
    for seed in (0, 42):
        for depth in (20, 24):
            np.random.seed(seed)
            S, Fs, Ds = [], [], []
            for _ in range(16):
                s, f, d = cube.scramble(depth, True)
                S.append(s), Fs.append(f), Ds.append(d)
            fx[f"scr_s{seed}_d{depth}_states"] = np.array(S)
            fx[f"scr_s{seed}_d{depth}_faces"] = np.array(Fs)
            fx[f"scr_s{seed}_d{depth}_dirs"] = np.array(Ds)

    # (4) multi_rotate on 4096 reachable states, both directions; 12-child expansion of 256 states
    rng = np.random.RandomState(1234)
    states = _reachable_states(cube, 4096, 30, 7)
    faces, dirs = rng.randint(0, 6, 4096), rng.randint(0, 2, 4096)
    fx["mr_in"], fx["mr_faces"], fx["mr_dirs"] = states, faces.astype(np.uint8), dirs.astype(np.uint8)
    fx["mr_out"] = cube.multi_rotate(states, faces, dirs)
    assert fx["mr_out"].dtype == np.int8 and set(np.unique(dirs)) == {0, 1}
    parents = states[:256]
    fx["ex_parents"] = parents
    fx["ex_children"] = cube.multi_rotate(np.repeat(parents, 12, axis=0), *cube.iter_actions(256))
    # single-state rotate agrees row-wise (reference's own differential test, tests/test_cube.py:94-101)
    for i in range(64):
        assert np.array_equal(cube.rotate(states[i], faces[i], dirs[i]), fx["mr_out"][i])

    # (5) multi_is_solved with solved rows at known positions
    batch = _reachable_states(cube, 1000, 3, 11)   # depth 3: a few may be solved by chance
    planted = np.array([0, 63, 64, 65, 127, 128, 511, 999])
    batch[planted] = cube.get_solved()
    fx["is_in"] = batch
    fx["is_out"] = cube.multi_is_solved(batch)
    assert fx["is_out"][planted].all()

    # (6) as_oh: index form for 256 states + dense rows
    oh = cube.as_oh(states[:256]).cpu().numpy()
    assert oh.shape == (256, 480) and oh.dtype == np.float32 and (oh.sum(1) == 20).all()
    fx["oh_in"] = states[:256]
    fx["oh_cols"] = np.nonzero(oh)[1].reshape(256, 20).astype(np.int16)
    fx["oh_dense_row0"] = oh[0]
    fx["oh_single"] = cube.as_oh(states[5]).cpu().numpy()
    assert fx["oh_single"].shape == (1, 480)

    # (7) sequence_scrambler(8, 20, with_solved in {True, False}) after seed(0)
    for ws in (True, False):
        np.random.seed(0)
        s, oh = cube.sequence_scrambler(8, 20, ws)
        fx[f"seq_ws{int(ws)}_states"] = s
        fx[f"seq_ws{int(ws)}_ohcols"] = np.nonzero(oh.cpu().numpy())[1].reshape(len(s), 20).astype(np.int16)

    # (8) action helpers
    fx["iter_actions_2"] = cube.iter_actions(2)
    f12, d12 = cube.indices_to_actions(np.arange(12))
    fx["i2a_faces"], fx["i2a_dirs"] = f12, d12
    fx["rev_actions"] = cube.rev_actions(np.arange(12))
    fx["rev_action_scalar"] = np.array([cube.rev_action(a) for a in range(12)])
    fx["action_space"] = np.array(cube.action_space)

    # (9) sticker nets of random states (as633) -- complements the literal nets in the reference's tests
    fx["as633_in"] = states[:32]
    fx["as633_out"] = np.array([cube.as633(s) for s in states[:32]])

    np.savez_compressed(os.path.join(OUT, "cube_golden.npz"), **fx)
    print("cube_golden.npz:", {k: v.shape for k, v in fx.items()})


def make_bfs():
    """BASELINE config #1: BFS on 10 depth-5 scrambles after set_seeds() (runeval.py defaults)."""
    from librubiks import cube
    from librubiks.utils import set_seeds
    from librubiks.solving.agents import BFS
    set_seeds()
    agent = BFS()
    states, lengths, seen, queues = [], [], [], []
    for _ in range(10):
        s, _, _ = cube.scramble(5, True)
        ok = agent.search(s, None, 10_000_000)
        assert ok
        states.append(s), lengths.append(len(agent.action_queue)), seen.append(len(agent))
        queues.append(list(agent.action_queue) + [-1] * (5 - len(agent.action_queue)))
    print("BFS lengths", lengths, "states seen", seen)
    np.savez_compressed(os.path.join(OUT, "bfs_golden.npz"), states=np.array(states),
                        lengths=np.array(lengths), seen=np.array(seen), queues=np.array(queues))


def make_bfs_cut():
    """BFS searches of the reference that END BY THE max_states TEST (and a few that just make it): pins len(agent) there."""
    from librubiks import cube
    from librubiks.solving.agents import BFS
    np.random.seed(77)
    agent = BFS()
    rows = []
    for depth in (3, 4, 5, 6, 7):
        s, _, _ = cube.scramble(depth, True)
        for cap in (1, 2, 13, 14, 100, 1000, 1234, 20000):
            ok = agent.search(s, None, cap)
            q = list(agent.action_queue)
            rows.append((s, cap, int(ok), len(agent), q + [-1] * (8 - len(q))))
    np.savez_compressed(os.path.join(OUT, "bfs_cut_golden.npz"), states=np.array([r[0] for r in rows]),
                        caps=np.array([r[1] for r in rows]), solved=np.array([r[2] for r in rows]),
                        seen=np.array([r[3] for r in rows]), queues=np.array([r[4] for r in rows]))
    print("bfs_cut_golden.npz:", len(rows), "searches,", sum(r[2] for r in rows), "solved")

def make_model():
    """
    The reference's own `Model` (librubiks/model.py:106-161, ResNet :250-264), imported: for every architecture x batchnorm
    the ordered (key, shape, dtype) list of `Model.create(ModelConfig(...)).state_dict()` under `torch.manual_seed(0)`, a
    SHA-256 per tensor of that state_dict (the parameters themselves are 50-200 MB), and the module's eval-mode outputs on
    the 256 golden `oh_in` states, in fp32 as the reference runs them and from the same module cast to float64.  For the
    batchnorm variants additionally: three train-mode forwards (256 states each) so that the BatchNorm statistics are not
    the initial ones, the statistics, and the eval-mode outputs behind them.
    """
    import hashlib
    import torch
    from librubiks import cube
    from librubiks.model import Model, ModelConfig
    g = np.load(os.path.join(OUT, "cube_golden.npz"))
    oh_eval = cube.as_oh(g["oh_in"]).cpu()
    oh_train = [cube.as_oh(g["mr_in"][256 * i:256 * (i + 1)]).cpu() for i in (1, 2, 3)]
    fx, meta = {}, {}

    def outputs(net, tag):
        net.eval()
        with torch.no_grad():
            p, v = net(oh_eval)
            net64 = Model.create(net.config).double()
            net64.load_state_dict({k: (t.double() if t.is_floating_point() else t) for k, t in net.state_dict().items()})
            net64.eval()
            p64, v64 = net64(oh_eval.double())
        assert p.dtype == torch.float32 and p.shape == (256, 12) and v.shape == (256, 1)
        fx[f"{tag}_p32"], fx[f"{tag}_v32"] = p.numpy(), v.numpy()
        fx[f"{tag}_p64"], fx[f"{tag}_v64"] = p64.numpy(), v64.numpy()

    for arch in ("fc_small", "fc_big", "res_small", "res_big"):
        for bn in (True, False):
            name = f"{arch}_bn{int(bn)}"
            torch.manual_seed(0)
            net = Model.create(ModelConfig(architecture=arch, batchnorm=bn))
            assert net.training                       # a fresh module is in train mode (agents call .eval(), agents.py:53)
            sd = net.state_dict()
            meta[name] = {"keys": [[k, list(t.shape), str(t.dtype)] for k, t in sd.items()],
                          "sha256": {k: hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest() for k, t in sd.items()},
                          "n_params": int(sum(p.numel() for p in net.parameters())),
                          "repr": repr(net)}
            outputs(net, f"{name}_fresh")
            if bn:
                net.train()
                with torch.no_grad():
                    for x in oh_train:
                        net(x)
                for k, t in net.state_dict().items():
                    if "running_" in k or "num_batches" in k:
                        fx[f"{name}_stat_{k}"] = t.numpy().copy()
                outputs(net, f"{name}_trained_stats")
            print(name, meta[name]["n_params"], "parameters,", len(sd), "tensors")
    fx["meta_json"] = np.array(json.dumps(meta))
    fx["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "model_golden.npz"), **fx)
    print("model_golden.npz:", len(fx), "arrays,", os.path.getsize(os.path.join(OUT, "model_golden.npz")), "bytes")


if __name__ == "__main__":
    what = sys.argv[1:] or ["cube", "bfs", "bfs_cut", "agents", "adi", "simple"]
    assert os.path.isdir("/root/reference"), "needs the mounted reference"
    if "cube" in what:
        make_cube()
    if "bfs" in what:
        make_bfs()
    if "bfs_cut" in what:
        make_bfs_cut()
    if "agents" in what:
        from make_golden_agents import make_agents
        make_agents()
    if "simple" in what:
        from make_golden_agents import make_simple_agents
        make_simple_agents()
    if "adi" in what:
        from make_golden_agents import make_adi
        make_adi()
    if "model" in what:
        make_model()
