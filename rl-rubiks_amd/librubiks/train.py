"""
Autodidactic-iteration training with the reference's `Train` interface (librubiks/train.py:18-412),
with the whole data-generation path resident on the GPU:

    sequence_scrambler (HIP) -> expand12 (HIP) -> is_solved (HIP) -> value net -> rc_adi_targets (HIP)

so the rollout's (games * depth * 12) substates never exist on the host (the reference builds them
in NumPy and ships a 377 MB one-hot matrix to the GPU every rollout at BASELINE config #4).
The optimisation loop is stock PyTorch, like the reference's.  Plots and TrainAnalysis are out of
scope.  Data-parallel training over several GPUs: every rank generates games / world_size games of
each rollout from its own NumPy stream and gradients are averaged over the ranks (`average_gradients`).
"""
from contextlib import contextmanager

import numpy as np
import torch
import torch.distributed as dist

from librubiks import _hip, cube, gpu, no_grad
from librubiks.cube.device import DeviceCubes
from librubiks.model import Model, make_inference_net, net_fingerprint
from librubiks.utils import NullLogger, TickTock

_FIX = {"paper": 0, "reward0": 0, "lapanfix": 1, "schultzfix": 2}


def average_gradients(net: torch.nn.Module):
    """Mean of the gradients over all ranks in one flat all_reduce (RCCL on GPUs); no-op for a single process.
    The blocking fallback; Train.train uses GradBuckets, which overlaps the collective with backward."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    offset = 0
    for g in grads:
        g.copy_(flat[offset:offset + g.numel()].view_as(g))
        offset += g.numel()


class GradBuckets:
    """
    Data-parallel gradient averaging overlapped with backward.  The parameters' `.grad` tensors are views into a few
    flat buckets (no gather / scatter copies); a bucket's exchange is launched asynchronously the moment
    backward has accumulated its last gradient, so it runs on RCCL's stream while autograd is still computing the
    gradients of the earlier layers.  Buckets are filled in the order backward produces gradients (last layer
    first); ~16 MB each keeps the per-call latency (~20 us) negligible while the first one starts after
    the two head layers, i.e. under the backward of the three large trunk GEMMs.

    exchange (the eight GPUs of a node are fully connected point to point, 7 xGMI links x ~150 GB/s each):
      "ring"    one all_reduce per bucket: RCCL's ring moves 2 (W-1)/W of a bucket over ONE link (~0.6 ms per step for the 50 MB
                of fp32 gradients of fc_small at W = 8);
      "direct"  reduce-scatter and all-gather over all links at once: an all_to_all_single hands rank j everybody's j-th shard
                (each rank sends (W-1)/W of the bucket, a W-th over each link), rank j adds them -- one fixed order, so every rank
                ends with bit-identical gradients -- and an all_gather_into_tensor returns the sums; the first half runs under
                backward, the second is issued in wait();
      "auto"    ring.  The direct form is OPT-IN (`exchange="direct"`): it has run on gloo (2, 3, 8 ranks) and on a one-rank RCCL
                group, never between GPUs, and it doubles the gradient buffers (`recvs`); it becomes the default on RCCL only once
                an 8-GPU run has shown it bit-identical across ranks and faster than the ring.
    Neither form has been timed on xGMI (no multi-GPU box was available to this build); both are tested on gloo.
    `wait()` = every bucket reduced and divided by the world size; `zero()` replaces optimizer.zero_grad().
    """

    def __init__(self, net: torch.nn.Module, bucket_bytes: int = 16 << 20, exchange: str = "auto"):
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if on else 1
        assert exchange in ("auto", "ring", "direct")
        if exchange == "auto":
            exchange = "ring"
        self.direct = exchange == "direct" and self.world > 1
        params = [p for p in net.parameters() if p.requires_grad]
        self.flats, self.bucket_of, self.sizes = [], {}, []
        group, nbytes = [], 0
        groups = []
        for p in reversed(params):                 # backward order
            group.append(p)
            nbytes += p.numel() * p.element_size()
            if nbytes >= bucket_bytes:
                groups.append(group)
                group, nbytes = [], 0
        if group:
            groups.append(group)
        for b, group in enumerate(groups):
            n = sum(p.numel() for p in group)
            n = -(-n // self.world) * self.world     # whole shards (the padding stays zero)
            flat = torch.zeros(n, dtype=group[0].dtype, device=group[0].device)
            off = 0
            for p in group:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self.bucket_of[p] = b
            self.flats.append(flat)
            self.sizes.append(len(group))
        self.recvs = [torch.empty_like(f) for f in self.flats] if self.direct else []
        self.left = list(self.sizes)
        self.works = []                             # (bucket, work handle of its first collective)
        self.hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in params]
        self.bytes = sum(f.numel() * f.element_size() for f in self.flats)

    def _launch(self, b: int):
        if self.direct:      # shard j of everybody's bucket -> rank j
            self.works.append((b, dist.all_to_all_single(self.recvs[b], self.flats[b], async_op=True)))
        else:
            self.works.append((b, dist.all_reduce(self.flats[b], op=dist.ReduceOp.SUM, async_op=True)))

    def _ready(self, p):
        b = self.bucket_of[p]
        self.left[b] -= 1
        if self.left[b] == 0 and self.world > 1:
            self._launch(b)

    def zero(self):
        for f in self.flats:
            f.zero_()
        self.left = list(self.sizes)

    def wait(self):
        """All buckets summed over the ranks and divided by the world size.  A bucket whose parameters did not ALL receive a
        gradient in this backward pass (a loss that uses one head only, a frozen or unused branch) never fired from the hooks:
        it is reduced here -- every rank reaches this point with the same set of such buckets, because which parameters get a
        gradient is a property of the graph, not of the data -- so no rank steps on un-summed gradients."""
        if self.world > 1:
            for b, left in enumerate(self.left):
                if left > 0:
                    self._launch(b)
                    self.left[b] = 0
        gathers = []
        for b, w in self.works:
            w.wait()
            if self.direct:   # this rank's shard of the sum (ranks added in one fixed order), then back to everybody
                shard = self.recvs[b].view(self.world, -1).sum(0)
                gathers.append((shard, dist.all_gather_into_tensor(self.flats[b], shard, async_op=True)))
        for _, g in gathers:      # (the shards live until their gathers have run)
            g.wait()
        self.works = []
        if self.world > 1:
            for f in self.flats:
                f.div_(self.world)

    def close(self):
        for h in self.hooks:
            h.remove()


def average_buffers(net: torch.nn.Module):
    """Mean of the floating-point buffers (BatchNorm running statistics) over all ranks: each rank normalises its own
    games, so without this the replicas' eval-mode networks drift apart although their weights are identical."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    bufs = [b for b in net.buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= dist.get_world_size()
    offset = 0
    for b in bufs:
        b.copy_(flat[offset:offset + b.numel()].view_as(b))
        offset += b.numel()


class Train:
    def __init__(self, rollouts: int, batch_size: int, rollout_games: int, rollout_depth: int, optim_fn,
                 alpha_update: float, lr: float, gamma: float, update_interval: int, agent, evaluator,
                 evaluation_interval: int, with_analysis: bool = False, tau: float = 1, reward_method: str = "lapanfix",
                 policy_criterion=torch.nn.CrossEntropyLoss, value_criterion=torch.nn.MSELoss, logger=NullLogger(),
                 adi_net_dtype=torch.float32):
        assert reward_method in _FIX, f"reward_method must be one of {sorted(_FIX)}"
        assert not with_analysis, "TrainAnalysis is out of scope of the MI355X build"
        self.rollouts, self.rollout_games, self.rollout_depth = rollouts, rollout_games, rollout_depth
        self.states_per_rollout = rollout_depth * rollout_games
        self.batch_size = batch_size or self.states_per_rollout
        self.reward_method, self.tau = reward_method, tau
        self.alpha_update, self.lr, self.gamma, self.update_interval = alpha_update, lr, gamma, update_interval
        self.optim = optim_fn
        self.policy_criterion = policy_criterion(reduction="none")
        self.value_criterion = value_criterion(reduction="none")
        self.agent, self.evaluator, self.log = agent, evaluator, logger
        self.adi_net_dtype = adi_net_dtype
        # evaluation schedule of the reference (train.py:63-73)
        if evaluation_interval:
            ev = np.arange(0, rollouts, evaluation_interval) - 1
            ev = ev[1:] if evaluation_interval == 1 else np.concatenate([[0], ev[1:]])
            if len(ev) == 0 or ev[-1] != rollouts - 1:
                ev = np.append(ev, rollouts - 1)
            self.evaluation_rollouts = ev
        else:
            self.evaluation_rollouts = np.array([])
        self.tt = TickTock()
        self.adi_chunk = 1 << 19   # substates per value-network call

    # ---- data generation (train.py:257-339), device resident -------------------------------------
    @no_grad
    def ADI_traindata(self, net, alpha: float):
        """
        (one-hot states float32[G*D, 480], policy targets int64[G*D], value targets float32[G*D],
        loss weights float32[G*D]), all on the GPU.
        """
        lib, st = _hip.lib(), _hip.stream_ptr()
        net.eval()
        G, D = self.rollout_games, self.rollout_depth
        states = cube.sequence_scrambler_device(G, D, with_solved=self.reward_method == "lapanfix")
        n = states.n
        kids, state_solved, kid_solved = states.expand12_flags()      # one launch: the children and both solved tests (train.py:285-296)
        kid_solved, state_solved = kid_solved.view(torch.uint8), state_solved.view(torch.uint8)
        values = torch.empty(12 * n, dtype=torch.float32, device=states.soa.device)
        engine = self._adi_engine(net)
        for lo in range(0, 12 * n, self.adi_chunk):   # chunked like the reference's adi_ff_batches (train.py:301-310)
            m = min(self.adi_chunk, 12 * n - lo)
            if engine is not None and engine.supports_cubes and lo % 16 == 0:
                values[lo:lo + m] = engine.value_cubes(kids, None, lo, m)   # a column window of the SoA, no copy
                continue
            part = DeviceCubes(kids.soa[:, lo:lo + ((m + 15) // 16) * 16].contiguous(), m) if (lo or m < 12 * n) else kids
            if engine is not None and engine.supports_cubes:
                values[lo:lo + m] = engine.value_cubes(part)
            else:
                values[lo:lo + m] = net(part.as_oh(torch.float32), policy=False, value=True).float().reshape(-1)
        policy_targets = torch.empty(n, dtype=torch.int64, device=values.device)
        value_targets = torch.empty(n, dtype=torch.float32, device=values.device)
        _hip.check(lib.rc_adi_targets(values.data_ptr(), kid_solved.data_ptr(), state_solved.data_ptr(), n, D,
                                      0.0 if self.reward_method == "reward0" else 1.0, _FIX[self.reward_method],
                                      policy_targets.data_ptr(), value_targets.data_ptr(), st), "rc_adi_targets")
        # loss weights (train.py:330-333): host arithmetic on a (G*D,) vector, independent of the states
        weighted = np.tile(1 / np.arange(1, D + 1), G)
        ws, us = weighted.sum(), len(weighted)
        loss_weights = ((1 - alpha) * weighted / ws + alpha * np.ones_like(weighted) / us) * (ws + us)
        return states.as_oh(torch.float32), policy_targets, value_targets, \
            torch.from_numpy(loss_weights).float().to(values.device)

    _engine_cache = None

    def _adi_engine(self, net):
        """The value network of the data generation: the torch module itself in fp32 (`adi_net_dtype=torch.float32`, the
        reference's arithmetic and the default), or one of the inference engines -- F32_SPLIT: fp32 accuracy on the f16 matrix
        cores, torch.bfloat16: the fast engine -- straight from the device-resident substates (no one-hot matrix).  Rebuilt when
        the generator's weights change (`net_fingerprint`)."""
        if self.adi_net_dtype == torch.float32 or not (isinstance(net, Model) and net.config.architecture.startswith("fc")):
            return None
        fp = net_fingerprint(net, self.adi_net_dtype)
        if self._engine_cache is None or self._engine_cache[0] != fp:
            self._engine_cache = (fp, make_inference_net(net, self.adi_net_dtype))
        return self._engine_cache[1]

    # ---- training loop (train.py:111-255) ----------------------------------------------------------
    def _shard_over_ranks(self):
        """Data-parallel ADI: with W ranks every rank generates rollout_games / W games per rollout from its own NumPy
        stream (the common seed + rank); gradients are averaged per step.  Single process: nothing changes."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1 or self._sharded:
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        self.rollout_games = -(-self.rollout_games // world)
        self.states_per_rollout = self.rollout_depth * self.rollout_games
        # The rank's own stream feeds the ADI scrambles only.  The global np.random stream stays COMMON to all ranks:
        # the evaluator draws its scrambles from it and shards them over the ranks (evaluation.py, sharding.py), which
        # is only a partition of one scramble set if every rank draws the same one.
        self._adi_stream = np.random.RandomState((int(np.random.get_state()[1][0]) + rank) % (2 ** 32)).get_state()
        self._sharded = True

    _sharded = False
    _adi_stream = None

    @contextmanager
    def _rank_stream(self):
        """Inside: np.random is this rank's private ADI stream; outside: the stream all ranks share."""
        if self._adi_stream is None:
            yield
            return
        common = np.random.get_state()
        np.random.set_state(self._adi_stream)
        try:
            yield
        finally:
            self._adi_stream = np.random.get_state()
            np.random.set_state(common)

    def train(self, net: Model):
        self.tt.reset()
        self.tt.tick()
        self._shard_over_ranks()
        best_solve, best_net = 0, net.clone()
        self.agent.net = net
        generator_net = net.clone()
        alpha = 1 if self.alpha_update == 1 else 0
        optimizer = self.optim(net.parameters(), lr=self.lr)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, 1, self.gamma)
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        buckets = GradBuckets(net) if distributed else None
        self.policy_losses, self.value_losses = np.zeros(self.rollouts), np.zeros(self.rollouts)
        self.train_losses, self.sol_percents = np.empty(self.rollouts), []
        for rollout in range(self.rollouts):
            generator_net = self._update_gen_net(generator_net, net) if self.tau != 1 else net
            self.tt.profile("ADI training data")
            with self._rank_stream():
                data, policy_targets, value_targets, loss_weights = self.ADI_traindata(generator_net, alpha)
            self.tt.end_profile("ADI training data")
            self.tt.profile("Training loop")
            net.train()
            batches = self._get_batches(len(data), self.batch_size)
            # the logged losses are summed on the device and read once per rollout: a host read per step (as the
            # reference does with float(...)) would stall the launch queue 2 x len(batches) times per rollout
            p_acc = torch.zeros((), dtype=torch.float64, device=data.device)
            v_acc = torch.zeros((), dtype=torch.float64, device=data.device)
            for batch in batches:
                buckets.zero() if buckets else optimizer.zero_grad()
                policy_pred, value_pred = net(data[batch], policy=True, value=True)
                policy_loss = self.policy_criterion(policy_pred, policy_targets[batch]) * loss_weights[batch]
                value_loss = self.value_criterion(value_pred.squeeze(1), value_targets[batch]) * loss_weights[batch]
                torch.mean(policy_loss + value_loss).backward()   # bucket all_reduces start inside, as gradients complete
                if buckets:
                    self.tt.profile("Gradient all-reduce wait")
                    buckets.wait()
                    self.tt.end_profile("Gradient all-reduce wait")
                optimizer.step()
                p_acc += policy_loss.detach().mean().double() / len(batches)
                v_acc += value_loss.detach().mean().double() / len(batches)
            average_buffers(net)
            self.policy_losses[rollout], self.value_losses[rollout] = float(p_acc), float(v_acc)
            self.train_losses[rollout] = self.policy_losses[rollout] + self.value_losses[rollout]
            self.tt.end_profile("Training loop")
            if rollout and self.update_interval and rollout % self.update_interval == 0:   # train.py:190-200
                if self.gamma != 1:
                    scheduler.step()
                if self.alpha_update and (alpha + self.alpha_update <= 1 or np.isclose(alpha + self.alpha_update, 1)):
                    alpha += self.alpha_update
                elif self.alpha_update and alpha < 1 and alpha + self.alpha_update > 1:
                    alpha = 1
            if rollout in self.evaluation_rollouts and self.evaluator is not None:
                net.eval()
                self.agent.net = net
                results, _, _ = self.evaluator.eval(self.agent)
                reward = float((results != -1).mean())
                self.sol_percents.append(reward)
                if reward > best_solve:
                    best_solve, best_net = reward, net.clone()
        if buckets:
            buckets.close()
        return net, best_net

    def _update_gen_net(self, generator_net: Model, net: Model):
        """generator <- tau * net + (1 - tau) * generator, over the whole state_dict (train.py:341-353)."""
        gen, new = generator_net.state_dict(), net.state_dict()
        for name, p in new.items():
            if gen[name].dtype.is_floating_point:
                gen[name].copy_(self.tau * p + (1 - self.tau) * gen[name])
            else:   # integer buffers (num_batches_tracked): same formula, truncated like the reference's copy_
                gen[name].copy_((self.tau * p + (1 - self.tau) * gen[name]).to(gen[name].dtype))
        return generator_net.to(gpu)

    @staticmethod
    def _get_batches(size: int, bsize: int):
        """Contiguous slices; the reference shuffles an index array it then never uses (train.py:400-410)."""
        n = int(np.ceil(size / bsize))
        return [slice(b * bsize, min(size, (b + 1) * bsize)) for b in range(n)]
