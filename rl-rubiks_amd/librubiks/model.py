"""
Policy/value network of the reference (librubiks/model.py:15-264) on PyTorch-ROCm.

`ModelConfig` / `Model` keep the reference's constructor arguments, module layout and file layout
(`model.pt` / `model-best.pt` state_dict + `config.json`), so checkpoints written by the reference
load here and vice versa: the state_dict keys are the same (`shared_net.0.weight`,
`shared_net.2.running_mean`, `policy_net.3.bias`, `shared_net.resblock0.layer1.weight`, ...).

`InferenceNet` is what the batched search agents call: the eval-mode network with every
BatchNorm folded into the following Linear, the two heads merged into shared GEMMs, weights in
bf16 (MFMA through hipBLASLt) or fp32, fixed-shape buffers so that a whole search iteration can be
captured into a HIP graph.  The forward is the only dense contraction on the hot path; everything
else is integer work in csrc/.
"""
import ctypes
import json
import os
import warnings
from time import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from librubiks import gpu
from librubiks.utils import NullLogger

OH_WIDTH = 480
N_ACTIONS = 12

_ARCHS = {   # reference model.py:17-21
    "fc_small": {"shared_sizes": [4096, 2048], "part_sizes": [512]},
    "fc_big": {"shared_sizes": [8192, 4096, 2048], "part_sizes": [1024, 512]},
    "res_small": {"shared_sizes": [4096, 1024], "part_sizes": [512], "res_blocks": 4, "res_size": 1024},
    "res_big": {"shared_sizes": [8192, 4096, 2048], "part_sizes": [1024, 512], "res_blocks": 6, "res_size": 2048},
}
_ACTIVATIONS = {"elu": nn.ELU, "relu": nn.ReLU}


class ModelConfig:
    def __init__(self, activation_function=None, batchnorm=True, architecture="fc_small", init="glorot",
                 is2024=True, **kwargs):   # unknown keys (e.g. `id`) are swallowed like the reference does (model.py:28)
        self.activation_function = activation_function if activation_function is not None else nn.ELU()
        self.batchnorm = batchnorm
        self.architecture = architecture + "_small" if architecture in ("fc", "res") else architecture   # model.py:52-56
        self.init = init
        self.is2024 = is2024
        self.id = hash(time())
        if self.architecture == "conv":
            raise NotImplementedError("the conv architecture needs the 6x8x6 representation (out of scope on MI355X)")
        if self.architecture not in _ARCHS:
            raise KeyError(f"Network architecture should be one of {sorted(_ARCHS)}, but '{architecture}' was given")
        if not is2024:
            raise NotImplementedError("only the 20x24 representation is implemented on MI355X")
        arch = _ARCHS[self.architecture]
        self.shared_sizes = list(arch["shared_sizes"])
        self.part_sizes = list(arch["part_sizes"])
        if self.architecture.startswith("res"):
            self.res_blocks, self.res_size = arch["res_blocks"], arch["res_size"]

    def as_json_dict(self):
        name = [k for k, cls in _ACTIVATIONS.items() if isinstance(self.activation_function, cls)][0]
        return {"activation_function": name, "batchnorm": self.batchnorm, "architecture": self.architecture,
                "init": self.init, "is2024": self.is2024, "id": self.id}

    @classmethod
    def from_json_dict(cls, conf: dict):
        conf = dict(conf)
        conf["activation_function"] = _ACTIVATIONS[conf["activation_function"]]()
        return cls(**conf)


class NonConvResBlock(nn.Module):
    """Two same-width Linear layers with a skip connection (reference model.py:221-247)."""

    def __init__(self, width: int, activation: nn.Module, with_batchnorm: bool):
        super().__init__()
        self.layer1, self.layer2 = nn.Linear(width, width), nn.Linear(width, width)
        self.activate = activation
        self.with_batchnorm = with_batchnorm
        if with_batchnorm:
            self.batchnorm1, self.batchnorm2 = nn.BatchNorm1d(width), nn.BatchNorm1d(width)

    def forward(self, x):
        y = self.layer1(x)
        if self.with_batchnorm:
            y = self.batchnorm1(y)
        y = self.layer2(self.activate(y))
        if self.with_batchnorm:
            y = self.batchnorm2(y)
        return self.activate(y + x)


class Model(nn.Module):
    """Shared trunk 480 -> shared_sizes, then a policy head (12 logits) and a value head (1)."""

    def __init__(self, config: ModelConfig, logger=NullLogger()):
        super().__init__()
        self.config, self.log = config, logger
        trunk_out = config.shared_sizes[-1]
        # Construction order = the reference's (model.py:117-129, 256-264): trunk, policy head, value head, THEN the residual
        # blocks appended to the trunk -- the order in which the initialisers draw from torch's generator, so that a seeded
        # `Model.create` has the reference's parameters bit for bit (tests/test_model.py, tests/golden/model_golden.npz).
        self.shared_net = nn.Sequential(*self._stack([OH_WIDTH, *config.shared_sizes], last_is_output=False))
        self.policy_net = nn.Sequential(*self._stack([trunk_out, *config.part_sizes, N_ACTIONS], last_is_output=True))
        self.value_net = nn.Sequential(*self._stack([trunk_out, *config.part_sizes, 1], last_is_output=True))
        if config.architecture.startswith("res"):
            assert trunk_out == config.res_size
            for i in range(config.res_blocks):
                self.shared_net.add_module(f"resblock{i}",
                                           NonConvResBlock(config.res_size, config.activation_function, config.batchnorm))

    @staticmethod
    def create(config: ModelConfig, logger=NullLogger()):
        return Model(config, logger).to(gpu)

    def _stack(self, widths, last_is_output: bool):
        """Linear -> activation -> BatchNorm1d per hidden layer (model.py:143-161); a bare Linear at an output."""
        layers = []
        for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
            lin = nn.Linear(fan_in, fan_out)
            if self.config.init == "glorot":
                nn.init.xavier_uniform_(lin.weight)
            elif self.config.init == "he":
                nn.init.kaiming_uniform_(lin.weight)
            else:
                nn.init.constant_(lin.weight, float(self.config.init))
            layers.append(lin)
            if not (last_is_output and i == len(widths) - 2):
                layers.append(self.config.activation_function)
                if self.config.batchnorm:
                    layers.append(nn.BatchNorm1d(fan_out))
        return layers

    def forward(self, x, policy=True, value=True):
        assert policy or value
        x = self.shared_net(x)
        out = []
        if policy:
            out.append(self.policy_net(x))
        if value:
            out.append(self.value_net(x))
        return out if len(out) > 1 else out[0]

    def clone(self):
        twin = Model.create(self.config)
        twin.load_state_dict({k: v.clone() for k, v in self.state_dict().items()})
        return twin

    def get_params(self):
        return torch.cat([x.float().flatten() for x in self.state_dict().values()]).clone()

    def save(self, save_dir: str, is_min=False):
        os.makedirs(save_dir, exist_ok=True)
        if is_min:
            torch.save(self.state_dict(), os.path.join(save_dir, "model-best.pt"))
            return
        torch.save(self.state_dict(), os.path.join(save_dir, "model.pt"))
        with open(os.path.join(save_dir, "config.json"), "w", encoding="utf-8") as f:
            json.dump(self.config.as_json_dict(), f, indent=4)

    @staticmethod
    def load(load_dir: str, logger=NullLogger(), load_best=False):
        with open(os.path.join(load_dir, "config.json"), encoding="utf-8") as f:
            config = ModelConfig.from_json_dict(json.load(f))
        path = os.path.join(load_dir, "model-best.pt" if load_best else "model.pt")
        if not os.path.exists(path):   # fall back like the reference (model.py:202-206)
            path = os.path.join(load_dir, "model.pt")
        model = Model.create(config, logger)
        model.load_state_dict(torch.load(path, map_location=gpu))
        return model.to(gpu)


# =================================================================================================
# Inference engine for the search agents
# =================================================================================================
def _bn_affine(bn: nn.BatchNorm1d):
    scale = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
    return scale, bn.bias.double() - bn.running_mean.double() * scale


class LayerOpts:
    """What a folded layer does besides y = act(W x + b): `save` = its input is the skip connection of a residual block,
    `add` = that input is added in front of the activation (reference model.py:221-247: activate(bn2(layer2(.)) + x)),
    `post` = (scale, shift) of an eval-mode BatchNorm BEHIND the activation that could not be folded into the next Linear
    because a skip connection reads its output as well."""
    __slots__ = ("save", "add", "post")

    def __init__(self, save=False, add=False, post=None):
        self.save, self.add, self.post = save, add, post


def _fold_stack(seq: nn.Sequential, carry=None):
    """
    [(W, b, activation)] of a Linear/act/BN stack in eval mode with each BatchNorm folded into the
    NEXT Linear; `carry` = (scale, shift) of a BatchNorm that precedes the stack.  Returns the list
    and the trailing BatchNorm's affine (or None) for whoever consumes the stack's output.
    NonConvResBlocks (the res_* architectures, reference model.py:221-264) become two entries each: Linear -> BN -> act
    folds the BatchNorm into ITS OWN Linear; the entries' LayerOpts (attribute `opts` of the returned list, one per entry)
    mark the skip connection, and a BatchNorm in front of the first block becomes a post-activation affine of the layer
    before it.
    """
    out, opts, mods, i = _Layers(), [], list(seq), 0
    out.opts = opts
    while i < len(mods):
        lin = mods[i]
        if isinstance(lin, NonConvResBlock):
            if carry is not None:   # the block's input feeds a Linear AND the skip: the BatchNorm in front of it is applied for real
                assert out, "a residual block cannot open the network behind a BatchNorm"
                opts[-1] = LayerOpts(opts[-1].save, opts[-1].add, (carry[0], carry[1]))
                carry = None
            for j, (layer, bn) in enumerate(((lin.layer1, getattr(lin, "batchnorm1", None)), (lin.layer2, getattr(lin, "batchnorm2", None)))):
                W, b = layer.weight.double(), layer.bias.double()
                if lin.with_batchnorm:
                    scale, shift = _bn_affine(bn)
                    W, b = W * scale[:, None], b * scale + shift
                out.append((W, b, lin.activate))
                opts.append(LayerOpts(save=j == 0, add=j == 1))
            i += 1
            continue
        assert isinstance(lin, nn.Linear), f"cannot fold {type(lin).__name__}"
        W, b = lin.weight.double(), lin.bias.double()
        if carry is not None:
            scale, shift = carry
            b = b + W @ shift
            W = W * scale[None, :]
            carry = None
        i += 1
        has_act = i < len(mods) and isinstance(mods[i], (nn.ELU, nn.ReLU))
        act = mods[i] if has_act else None
        i += int(has_act)
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
            carry = _bn_affine(mods[i])
            i += 1
        out.append((W, b, act))
        opts.append(LayerOpts())
    return out, carry


class _Layers(list):
    """A list of folded layers with their LayerOpts in `.opts` (same length)."""
    opts = None


class InferenceNet:
    """
    Eval-mode forward of an fc_* `Model` as a chain of fused GEMMs:
        x -> [Linear+act]* (trunk) -> [Linear+act] (both heads' first layers side by side) -> ... -> [13 outputs]
    Returns (policy logits float32[n,12], values float32[n]).  Any other torch module (ResNet models,
    test stand-ins) goes through `GenericNet`, which simply calls it.
    """

    def __init__(self, model: Model, dtype=torch.bfloat16, device=None, first_layer_table: str = "auto"):
        assert isinstance(model, Model) and model.config.architecture.split("_")[0] in ("fc", "res")
        device = device or next(model.parameters()).device
        was_training = model.training
        model.eval()
        with torch.no_grad():
            trunk, carry = _fold_stack(model.shared_net)
            pol, pc = _fold_stack(model.policy_net, carry)
            val, vc = _fold_stack(model.value_net, carry)
            assert pc is None and vc is None and len(pol) == len(val)
            layers, opts = list(trunk), list(trunk.opts)
            for d, ((Wp, bp, ap), (Wv, bv, av)) in enumerate(zip(pol, val)):
                if d == 0:   # both heads read the trunk output: stack the rows
                    W, b = torch.cat([Wp, Wv]), torch.cat([bp, bv])
                else:        # afterwards the heads are independent: block diagonal
                    W = torch.block_diag(Wp, Wv)
                    b = torch.cat([bp, bv])
                layers.append((W, b, ap))
                opts.append(LayerOpts())
            cast = lambda ls: [(W.to(device=device, dtype=dtype).contiguous(), b.to(device=device, dtype=dtype), act)  # noqa: E731
                               for W, b, act in ls]
            self.layers = cast(layers)
            self.value_layers = cast(list(trunk) + list(val))   # A* needs the value head only (agents.py:380)
            # skip connections / post-activation affines of the residual architectures, per layer tuple (the lists above are
            # sliced freely by callers; the tuples are not copied)
            self._opts = {}
            for ls, os_ in ((self.layers, opts), (self.value_layers, list(trunk.opts) + [LayerOpts()] * len(val))):
                for layer, o in zip(ls, os_):
                    post = None if o.post is None else tuple(t.to(device=device, dtype=torch.float32).contiguous() for t in o.post)
                    self._opts[id(layer)] = LayerOpts(o.save, o.add, post)
            self.residual = any(o.save for o in opts)
        model.train(was_training)
        self.dtype, self.device = dtype, device
        self.flops_per_state = 2 * sum(W.shape[0] * W.shape[1] for W, _, _ in self.layers)
        # Input layer fused with the one-hot encoding on the matrix cores (csrc/rubiks_net.hip): needs bf16 and H % 128 == 0.
        # The Linear weight as stored, in IEEE half when every weight fits its range (11 mantissa bits, closer to fp32 than bf16;
        # v_mfma_f32_32x32x16_f16), in bf16 otherwise (checkpoints without BatchNorm / with `he` init / early in training).
        W1, b1, act1 = self.layers[0]
        self._fused_first = None
        assert first_layer_table in ("auto", "onehot"), first_layer_table
        if dtype == torch.bfloat16 and W1.is_cuda and W1.shape[0] % 128 == 0 and W1.shape[1] == OH_WIDTH \
                and first_layer_table != "onehot":
            code = 0 if act1 is None else 1 if isinstance(act1, nn.ReLU) else 2
            W1_full = layers[0][0].to(device)
            half_ok = bool(torch.isfinite(W1_full).all()) and float(W1_full.abs().max()) < 3.0e4
            table = W1_full.to(torch.float16 if half_ok else torch.bfloat16).contiguous()
            self._fused_first = (table, layers[0][1].to(device).float().contiguous(), code,
                                 float(getattr(act1, "alpha", 1.0)), W1.shape[0], half_ok)

    @property
    def input_dtype(self):
        return self.dtype

    @torch.no_grad()
    def __call__(self, oh: torch.Tensor):
        out = self._run(self.layers, oh).float()
        return out[:, :N_ACTIONS], out[:, N_ACTIONS]

    @property
    def supports_cubes(self) -> bool:
        return self._fused_first is not None

    def workspace(self, rows: int):
        """Buffer for the fused input layer's output (`x1` of the *_cubes methods), or None."""
        if self._fused_first is None:
            return None
        return torch.empty((rows, self._fused_first[4]), dtype=torch.bfloat16, device=self.device)

    @torch.no_grad()
    def first_layer(self, cubes, out: torch.Tensor = None, lo: int = 0, n: int = None) -> torch.Tensor:
        """act(Linear(as_oh(cubes[lo:lo+n]))) as one HIP kernel: bf16 [n, H] without a one-hot matrix (lo % 16 == 0)."""
        from librubiks import _hip
        w1t, b1, code, alpha, H, is_f16 = self._fused_first
        if lo or n is not None:   # a column window of the SoA: the same planes, shifted base pointer
            n = cubes.n - lo if n is None else n
            assert lo % 16 == 0 and 0 <= lo and lo + n <= cubes.n
            cubes = _CubeWindow(cubes.soa.data_ptr() + lo, n, cubes.stride)
        if out is None:
            out = torch.empty((cubes.n, H), dtype=torch.bfloat16, device=w1t.device)
        _hip.check(_hip.lib().rc_first_layer_mfma_bf16(_soa_ptr(cubes), cubes.n, cubes.stride, w1t.data_ptr(),
                                                       b1.data_ptr(), out.data_ptr(), H, code, alpha, int(is_f16),
                                                       _hip.stream_ptr()), "rc_first_layer_mfma_bf16")
        return out

    def _fused_head_ok(self) -> bool:
        """rc_head_bf16 applies the last hidden activation and the 1024 -> 13 output layer in one pass."""
        W_out, _, act_out = self.layers[-1]
        return (self.dtype == torch.bfloat16 and len(self.layers) >= 3 and act_out is None
                and tuple(W_out.shape) == (N_ACTIONS + 1, 1024) and self.layers[-2][2] is not None and W_out.is_cuda)

    @torch.no_grad()
    def head_cubes(self, cubes, x1: torch.Tensor = None) -> torch.Tensor:
        """
        Output of the merged heads per row: 12 policy logits, then the value.  [n, 16] float32 when the fused head
        kernel applies (last ELU + output layer in one pass over the raw 1024-wide activations), else the head
        GEMM's [n, 13] tensor in the engine's dtype.
        """
        x = self.first_layer(cubes, x1)
        if not self._fused_head_ok():
            return self._run(self.layers[1:], x)
        x = self._run(self.layers[1:-2], x)
        W3, b3, _ = self.layers[-2]
        return self.head_from_raw(torch.addmm(b3, x, W3.t()))   # pre-activation of the last hidden layer

    @torch.no_grad()
    def head_from_raw(self, raw: torch.Tensor) -> torch.Tensor:
        from librubiks import _hip
        act3 = self.layers[-2][2]
        W4, b4, _ = self.layers[-1]
        if getattr(self, "_b4_f32", None) is None:
            self._b4_f32 = b4.float().contiguous()
        out = torch.empty((raw.shape[0], 16), dtype=torch.float32, device=raw.device)
        code = 1 if isinstance(act3, nn.ReLU) else 2
        _hip.check(_hip.lib().rc_head_bf16(raw.data_ptr(), raw.shape[0], raw.shape[1], W4.data_ptr(), self._b4_f32.data_ptr(),
                                           N_ACTIONS + 1, out.data_ptr(), code, float(getattr(act3, "alpha", 1.0)),
                                           _hip.stream_ptr()), "rc_head_bf16")
        return out

    @torch.no_grad()
    def forward_cubes(self, cubes, x1: torch.Tensor = None):
        """(policy logits, values) straight from device-resident cube states."""
        out = self._run(self.layers[1:], self.first_layer(cubes, x1)).float()
        return out[:, :N_ACTIONS], out[:, N_ACTIONS]

    @torch.no_grad()
    def value_cubes(self, cubes, x1: torch.Tensor = None, lo: int = 0, n: int = None) -> torch.Tensor:
        """Value head only, float32[n], straight from device-resident cube states (optionally the window lo..lo+n)."""
        return self._run(self.value_layers[1:], self.first_layer(cubes, x1, lo, n)).float().reshape(-1)

    # bf16 hidden layers with an activation as one kernel (rc_gemm_layer_bf16: bias, skip connection, activation fused) where its
    # 352 x 256 tiles fill the chip: 0.168 ms against 0.195 ms for hipBLASLt + the activation pass at 11 264 x 4096 x 2048; narrower
    # layers and smaller batches stay with the library.  (Which kernel a row runs on depends on how many rows share its launch --
    # as it does for the library's own choice of tiles: a row's bf16 result is not bit-identical across batch shapes.)
    fused_hidden = True

    def _run(self, layers, x):
        skip = None
        for layer in layers:
            W, b, act = layer
            o = self._opts.get(id(layer)) or _PLAIN
            if o.save:
                skip = x
            if (act is not None and self.fused_hidden and x.dtype == torch.bfloat16 and x.is_cuda and x.is_contiguous()
                    and W.shape[1] % 64 == 0 and W.shape[0] % 256 == 0 and -(-x.shape[0] // 352) * (W.shape[0] // 256) >= 192):
                # 352 x 256 tiles on >= 3/4 of the CUs: measured 0.186 ms against 0.200 ms for hipBLASLt + the activation pass at
                # 11 264 x 4096 x 2048 (round 3 probe); layers without activation and narrow ones stay with the library
                if getattr(b, "_f32", None) is None:
                    b._f32 = b.float().contiguous()
                out = torch.empty((x.shape[0], W.shape[0]), dtype=torch.bfloat16, device=x.device)
                _layer_call("rc_gemm_layer_bf16", a=x, w=W, bias=b._f32, residual=skip if o.add else None,
                            post_scale=o.post[0] if o.post else None, post_shift=o.post[1] if o.post else None, out_bf16=out,
                            n_rows=x.shape[0], n_out=W.shape[0], k=W.shape[1], activation=1 if isinstance(act, nn.ReLU) else 2,
                            alpha=float(getattr(act, "alpha", 1.0)), tile=1, k_splits=1)
                x = out
                continue
            x = torch.addmm(b, x, W.t())
            if o.add:
                x += skip
            if act is not None:
                x = _activate_(x, act)
            if o.post is not None:
                x = (x * o.post[0].to(x.dtype) + o.post[1].to(x.dtype)) if x.dtype != torch.float32 else torch.addcmul(o.post[1], x, o.post[0])
        return x

    @torch.no_grad()
    def value(self, oh: torch.Tensor) -> torch.Tensor:
        """Value head only, float32[n]."""
        return self._run(self.value_layers, oh).float().reshape(-1)


_PLAIN = LayerOpts()
SPLIT_SCALE = 2.0 ** 11
F32_SPLIT = "f32_split"   # `net_dtype` value selecting SplitF32Net
# ... and SplitF32Net(deterministic=True): ONE layer plan whatever the row count, so that a state's outputs do not depend on which
# other states share its launch -- a game searched alone, in a batch, on fewer slots or in a narrowed forest builds the same tree bit
# for bit (agents: `MCTS(..., deterministic=True)`).  Costs throughput at both ends of the row-count range.
F32_SPLIT_DET = "f32_split_deterministic"
HALF_MAX = 65504.0


class SplitRangeError(ValueError):
    """The network does not fit the f16x3 split format (a folded weight beyond IEEE half's range)."""


class _SplitLayer(ctypes.Structure):   # mirrors rc_split_layer_t (include/rubiks_hip.h)
    _fields_ = [(n, ctypes.c_void_p) for n in ("a", "w", "bias", "residual", "post_scale", "post_shift", "out_hi_lo", "out_f32",
                                               "out_partials", "out_bf16")] + \
               [(n, ctypes.c_size_t) for n in ("n_rows", "n_out", "k")] + \
               [("activation", ctypes.c_int), ("alpha", ctypes.c_float), ("tile", ctypes.c_int), ("k_splits", ctypes.c_int),
                ("range_flag", ctypes.c_void_p), ("products", ctypes.c_int)]


def _ptr(t):
    return None if t is None else t.data_ptr()


_MM_OUT = [True]


def _mm_f32(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor):
    """out = a @ b with f16 operands and fp32 accumulation / output (hipBLASLt), into a slice of a partials buffer."""
    if _MM_OUT[0]:
        try:
            return torch.mm(a, b, out_dtype=torch.float32, out=out)
        except (RuntimeError, TypeError, NotImplementedError):
            _MM_OUT[0] = False   # this torch has no mm.dtype_out: one extra copy
    out.copy_(torch.mm(a, b, out_dtype=torch.float32))
    return out


def _layer_call(fn_name: str, **kw):
    """One rc_split_layer_t request (tensors by keyword, see the header) on the current stream."""
    from librubiks import _hip
    lib = _hip.lib()
    if lib.rc_split_layer_struct_bytes() != ctypes.sizeof(_SplitLayer):
        raise _hip.RubiksHipError("rc_split_layer_t differs between librubiks_hip.so and librubiks/model.py: rebuild the library")
    L = _SplitLayer()
    for name, v in kw.items():
        setattr(L, name, _ptr(v) if isinstance(v, torch.Tensor) or v is None else v)
    _hip.check(getattr(lib, fn_name)(ctypes.byref(L), _hip.stream_ptr()), fn_name)


class SplitF32Net:
    """
    The eval-mode forward of an fc_* `Model` at fp32 accuracy on the f16 matrix cores (MI355X has no TF32 and its fp32
    MFMA runs at 1/16 of the f16 rate).  Every float travels as two IEEE halves, x = hi + lo * 2^-11 (22 significant
    bits), and a layer is three f16 MFMA products with fp32 accumulation,
        y = act(hi_x W_hi^T + 2^-11 (hi_x W_lo^T + lo_x W_hi^T) + b),
    run by the own MFMA kernel (csrc/rubiks_gemm.hip): one launch per layer with the epilogue fused where whole-K tiles fill the chip,
    else with the K loop cut into chunks + a reduce kernel (`_layer_plan`); shapes the kernel does not take fall back to two library
    GEMMs.  The input layer is the sum of the 20 rows of W^T a state's cubies select, in fp32 straight from the cube states
    (rc_first_layer_gather_f16).  BatchNorm is
    folded in float64 as in InferenceNet, the two heads are merged, the 13-wide output layer runs in plain fp32.
    Against the float64 forward the error is BELOW that of the fp32 GEMM chain (tests/test_net_gpu.py), at ~2.5x its
    speed; it is the reference-precision engine of bench.py.  Same interface as InferenceNet.
    """
    dtype = F32_SPLIT
    input_dtype = torch.float32
    supports_cubes = True

    DET_CHUNKS = 4    # deterministic mode: every hidden layer's K loop is cut into this many chunks (2 where 3 k / 64 is not a multiple of 4)

    def __init__(self, model: Model, device=None, deterministic: bool = False):
        ref = InferenceNet(model, dtype=torch.float64, device=device, first_layer_table="onehot")
        self.deterministic = bool(deterministic)
        if self.deterministic:
            self.dtype = F32_SPLIT_DET
        self.device = ref.device
        self._opts = {}
        self.layers, self.value_layers = self._split(ref.layers, ref._opts), self._split(ref.value_layers, ref._opts)
        self.residual = ref.residual
        self.flops_per_state = ref.flops_per_state
        self.n_out = ref.layers[-1][0].shape[0]
        # Hidden activations travel as IEEE halves hi + lo 2^-11: a value beyond +-65504 (ELU is unbounded above; nets without
        # BatchNorm, `he` initialisation, the first batches of training) would become hi = inf.  Every kernel that writes the
        # split format ORs this device flag when that happens; the agents read it when they collect results (`overflowed`) and
        # repeat the search on `fallback()`, the fp32 GEMM chain -- the answer is never silently wrong.
        self.range_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._model, self._fallback = model, None
        self._zeros = {layer[1].shape[0]: torch.zeros(layer[1].shape[0], dtype=torch.float32, device=self.device)
                       for layer in self.layers + self.value_layers if layer[0] == "hid"}
        if self.deterministic:
            self._check_deterministic()

    def _check_deterministic(self):
        """Deterministic mode promises ONE summation order per output, whatever the batch: every layer must run on a kernel whose
        order does not depend on the row count.  Checked here, when the engine is built -- not in the middle of a search or of a
        graph capture -- and there is no silent way out: a network that does not fit raises, and `fallback()` (library GEMMs, which
        choose their kernels by batch shape) is refused."""
        for layers in (self.layers, self.value_layers):
            H = layers[0][5].shape[0]
            if not self.fused_input or H % 64 or len(layers) < 3:
                raise SplitRangeError(f"deterministic mode: the input layer ({H} wide, {len(layers)} layers) does not run on the fused one-hot MFMA kernel "
                                      "(width % 64 == 0, at least one hidden layer behind it); the library GEMM it would fall back to chooses its "
                                      "kernel by batch shape")
            for i, layer in enumerate(layers):
                if layer[0] == "hid":
                    self._layer_plan(352, layers, i)          # raises SplitRangeError for a shape the K-cut kernel does not take
            last = layers[-1]
            w = layers[-2][1].shape[0]
            if not (self.fused_head and last[0] == "f32" and last[1].shape[0] <= 16 and w in (512, 1024)):
                raise SplitRangeError(f"deterministic mode: the output layer ({w} -> {last[1].shape[0]}) does not run on rc_head_split_f32 (512 or 1 024 wide, "
                                      "at most 16 outputs); torch.addmm would choose its kernel by batch shape")

    def overflowed(self) -> bool:
        """True if an activation left half range since the last call (reads and clears the device flag: synchronises)."""
        hit = bool(self.range_flag.item())
        if hit:
            self.range_flag.zero_()
        return hit

    def fallback(self) -> "InferenceNet":
        """The same network on the fp32 MFMA GEMM chain (1/16 of the f16 rate, no range limit).  Not in deterministic mode: the
        library GEMMs choose their kernels by batch shape, so the promise would be broken without anybody noticing."""
        if self.deterministic:
            raise SplitRangeError("deterministic mode: an activation left IEEE half's range and the fp32 GEMM chain that would take over is not "
                                  "bit-reproducible across batch shapes; search these weights with deterministic=False")
        if self._fallback is None:
            self._fallback = InferenceNet(self._model, dtype=torch.float32, device=self.device)
        return self._fallback

    def _split(self, layers, ref_opts):
        out = []
        for i, ref_layer in enumerate(layers):
            W, b, act = ref_layer
            W, b = W.double(), b.double()
            if i == len(layers) - 1:
                assert act is None
                out.append(("f32", W.float().contiguous(), b.float().contiguous()))
                continue
            if not (bool(torch.isfinite(W).all()) and float(W.abs().max()) < 3.0e4):
                raise SplitRangeError(f"layer {i}: folded weights up to {float(W.abs().max()):.3g} do not fit IEEE half")
            assert W.shape[0] % 8 == 0
            hi = W.half()
            lo = ((W - hi.double()) * SPLIT_SCALE).half()
            code = 0 if act is None else 1 if isinstance(act, nn.ReLU) else 2
            alpha = float(getattr(act, "alpha", 1.0))
            if i == 0:    # one-hot input: x_lo = 0, so y = oh W_hi^T + 2^-11 oh W_lo^T = [oh, oh 2^-11] [W_hi | W_lo]^T
                out.append(("in", torch.cat([hi, lo], 1).contiguous(), b.float().contiguous(), code, alpha, hi.contiguous(), lo.contiguous(),
                            W.t().float().contiguous()))   # ... and W^T [480][H] in fp32: the rows rc_first_layer_gather_f16 adds up
            else:
                out.append(("hid", hi.contiguous(), torch.cat([lo, hi], 1).contiguous(), b.float().contiguous(), code, alpha,
                            torch.cat([lo, hi, hi], 1).contiguous()))   # [W_lo | W_hi | W_hi]: the operand of rc_split_gemm_f16
            o = ref_opts.get(id(ref_layer)) or _PLAIN
            self._opts[id(out[-1])] = LayerOpts(o.save, o.add, None if o.post is None else tuple(t.float().contiguous() for t in o.post))
        return out

    fused_hidden = True   # hidden layers as one kernel each (rc_split_gemm_f16) where its tile fills the chip
    fused_head = True     # last activation + output layer in one pass (rc_head_split_f32) behind a hidden layer that came as partials

    def _layer_plan(self, rows: int, layers, i: int):
        """How hidden layer i runs on `rows` rows: 'fused' (one rc_split_layer_f16 launch, whole K per workgroup), ('cut', tile, chunks)
        (the same kernel with its K loop cut into chunks + rc_split_reduce_f16), or 'library' (two hipBLASLt GEMMs + reduce / the fused head)."""
        Wh = layers[i][1]
        N, K = Wh.shape
        if self.deterministic:
            # one summation order for every launch: the K-cut form with a FIXED chunk count (partials summed in order by
            # rc_split_reduce_f16).  A row's partials do not depend on the tile or on the other rows (every tile walks its chunk of K in
            # the same order), so the tile may still follow the row count: 352 x 256 once that fills a quarter of the chip.
            steps = 3 * K // 64
            chunks = self.DET_CHUNKS if steps % self.DET_CHUNKS == 0 and steps // self.DET_CHUNKS >= 2 else 2
            if K % 64 or N % 128 or steps % chunks or steps // chunks < 2:
                raise SplitRangeError(f"deterministic mode needs layers the own kernel takes (k % 128 == 0, n_out % 128 == 0), not {K} -> {N}")
            wide = N % 256 == 0 and -(-rows // 352) * (N // 256) * chunks >= 64
            return ("cut", 1 if wide else 3, chunks)
        if self.fused_hidden and self._fused_tile(rows, N, K):
            return "fused"
        cut = self._k_split(rows, N, K) if self.fused_hidden else None
        return ("cut",) + cut if cut else "library"

    @staticmethod
    def _k_split(rows: int, n_out: int, k: int):
        """(tile, chunks) of rc_split_layer_f16 with out_partials for a layer too small to fill the chip with whole-K tiles: the K loop
        (3 k / 64 steps) cut into as many chunks as bring the launch to at most one workgroup per CU, 352 x 128 tiles (tile 3) for
        few rows or shallow layers (352 x 64, tile 7, where that would leave chunks of under six steps), 352 x 256 (tile 1) otherwise.
        None when no cut applies.  Measured against the two library GEMMs in profiles/r3_skinny_split_probe.txt (the library runs one 64 x 64 tile per workgroup over the whole K: latency-bound)."""
        if k % 64:
            return None
        steps, best = 3 * k // 64, None
        for tile, cols in ((3, 128), (1, 256)):
            if n_out % cols:
                continue
            base = -(-rows // 352) * (n_out // cols)
            chunks = max((c for c in range(2, 33) if steps % c == 0 and steps // c >= 2 and base * c <= 256), default=0)
            if chunks:
                key = (base * chunks, cols if rows * k >= 2048 * 4096 else -cols)   # wide tiles pay once A is large: deep layers, many rows
                if best is None or key > best[0]:
                    best = (key, (tile, chunks))
        if best is None or best[0][0] < 128:
            return None
        tile, chunks = best[1]
        if tile == 3 and steps // chunks < 6 and n_out % 64 == 0:
            # chunks of fewer than six K-steps are mostly pipeline fill: 352 x 64 tiles (tile 7) reach one workgroup per CU with half
            # the chunks -- chunks twice as deep, half the partials (2048 -> 1024 at 352 rows: 23.2 against 27.5 us with the reduce,
            # profiles/r6_kcut_tile7.txt; with twelve or more steps per chunk the narrow tile's activation re-reads cost more than that)
            base = -(-rows // 352) * (n_out // 64)
            c7 = max((c for c in range(2, 33) if steps % c == 0 and base * c <= 256), default=0)
            if c7 and steps // c7 >= 6 and base * c7 >= 128:
                return (7, c7)
        return best[1]

    @staticmethod
    def _fused_tile(rows: int, n_out: int, k: int) -> int:
        """Tile of rc_split_gemm_f16 for this layer, 0 = keep the two library GEMMs + rc_split_act_f16."""
        if k % 64 or n_out % 256:
            return 0
        # 352 x 256 tiles, one per CU: more than 128 of them (with 128 or fewer the K loop is cut in two instead: twice the
        # workgroups still fit one round; 160 tiles cut in two would take two rounds, 0.38 ms against 0.33 ms at 7 040 rows)
        return 1 if -(-rows // 352) * (n_out // 256) > 128 else 0

    def workspace(self, rows: int):
        return None

    # ---- operands ---------------------------------------------------------------------------------------------
    def _input_from_cubes(self, cubes, lo: int = 0, n: int = None) -> torch.Tensor:
        from librubiks import _hip
        if lo or n is not None:
            n = cubes.n - lo if n is None else n
            assert lo % 16 == 0 and 0 <= lo and lo + n <= cubes.n
            cubes = _CubeWindow(cubes.soa.data_ptr() + lo, n, cubes.stride)
        a = torch.empty((cubes.n, 2 * OH_WIDTH), dtype=torch.float16, device=self.device)
        _hip.check(_hip.lib().rc_oh_split_f16(_soa_ptr(cubes), cubes.n, cubes.stride, a.data_ptr(), _hip.stream_ptr()),
                   "rc_oh_split_f16")
        return a

    @staticmethod
    def _input_from_oh(oh: torch.Tensor) -> torch.Tensor:
        oh = oh.float()
        return torch.cat([oh, oh * (1.0 / SPLIT_SCALE)], 1).half()   # exact: entries are 0, 1 and 2^-11

    def _act(self, part: torch.Tensor, n_corr: int, bias: torch.Tensor, code: int, alpha: float, split: bool, skip=None, post=None) -> torch.Tensor:
        """post * act(sum of the partial products + bias + skip) + post' as [hi | lo] halves (split) or as fp32.  part: [P, n, w] fp32, the
        first n_corr of them correction products (still scaled by 2^11); skip: [n, 2 w] halves hi | lo."""
        from librubiks import _hip
        P, n, w = part.shape
        out = torch.empty((n, 2 * w), dtype=torch.float16, device=part.device) if split else torch.empty((n, w), dtype=torch.float32, device=part.device)
        _hip.check(_hip.lib().rc_split_reduce_f16(part.data_ptr(), n * w, P, n_corr, n, w, bias.data_ptr(), _ptr(skip), code, alpha,
                                                  _ptr(post[0]) if post else None, _ptr(post[1]) if post else None,
                                                  out.data_ptr() if split else None, None if split else out.data_ptr(),
                                                  self.range_flag.data_ptr(), _hip.stream_ptr()), "rc_split_reduce_f16")
        return out

    def _zero_bias(self, w: int) -> torch.Tensor:
        return self._zeros[w]   # made in __init__: never allocated (and filled) inside a graph capture

    fused_input = True   # the input layer as one kernel from the cube states when shapes allow ...
    gather_input = True  # ... rc_first_layer_gather_f16 (sum of 20 fp32 rows of W^T per state); False: rc_first_layer_split_f16 (one-hot MFMA, hi / lo tables)

    def _first_from_cubes(self, cubes, layers, lo: int = 0, n: int = None):
        """[hi | lo] activations of the input layer straight from device cubes, or None if the fused kernel does not apply."""
        from librubiks import _hip
        _, B, b, code, alpha, Wh, Wl, Wrows = layers[0]
        H = Wh.shape[0]
        if not self.fused_input or H % 64 or len(layers) < 3:
            return None
        if lo or n is not None:
            n = cubes.n - lo if n is None else n
            assert lo % 16 == 0 and 0 <= lo and lo + n <= cubes.n
            cubes = _CubeWindow(cubes.soa.data_ptr() + lo, n, cubes.stride)
        out = torch.empty((cubes.n, 2 * H), dtype=torch.float16, device=self.device)
        if self.gather_input:   # the one-hot row times W as the sum of 20 rows of W^T, in fp32 (no matrix cores: 20 additions per output)
            _hip.check(_hip.lib().rc_first_layer_gather_f16(_soa_ptr(cubes), cubes.n, cubes.stride, Wrows.data_ptr(), b.data_ptr(), out.data_ptr(), H,
                                                            code, alpha, self.range_flag.data_ptr(), _hip.stream_ptr()), "rc_first_layer_gather_f16")
            return out
        _hip.check(_hip.lib().rc_first_layer_split_flag_f16(_soa_ptr(cubes), cubes.n, cubes.stride, Wh.data_ptr(), Wl.data_ptr(), b.data_ptr(),
                                                            out.data_ptr(), H, code, alpha, self.range_flag.data_ptr(), _hip.stream_ptr()),
                   "rc_first_layer_split_flag_f16")
        return out

    def _forward_cubes(self, cubes, layers, lo: int = 0, n: int = None) -> torch.Tensor:
        a = self._first_from_cubes(cubes, layers, lo, n)
        if a is None:
            return self._forward(self._input_from_cubes(cubes, lo, n), layers)
        return self._forward(a, layers, first=1)

    @torch.no_grad()
    def _forward(self, a: torch.Tensor, layers, first: int = 0) -> torch.Tensor:
        """a: [n, 960] half operand of the input layer (or, with first = 1, its [hi | lo] output) -> fp32 [n, n_out]."""
        from librubiks import _hip
        skip = None
        for i, layer in enumerate(layers):
            if i < first:
                continue
            last_hidden = i == len(layers) - 2
            n = a.shape[0]
            o = self._opts.get(id(layer)) or _PLAIN
            if o.save:
                skip = a
            res, post = (skip if o.add else None), o.post
            if layer[0] == "in":
                _, B, b, code, alpha = layer[:5]
                part = torch.empty((1, n, B.shape[0]), dtype=torch.float32, device=a.device)
                _mm_f32(a, B.t(), part[0])
                n_corr = 0
            elif layer[0] == "hid":
                _, Wh, B2, b, code, alpha, W3 = layer
                K, w = Wh.shape[1], Wh.shape[0]
                plan = self._layer_plan(n, layers, i)
                if plan == "fused":   # one kernel: three products, bias, activation, re-split (or fp32 in front of the output layer)
                    out = torch.empty((n, w if last_hidden else 2 * w), dtype=torch.float32 if last_hidden else torch.float16, device=a.device)
                    _layer_call("rc_split_layer_f16", a=a, w=W3, bias=b, residual=res, post_scale=post[0] if post else None,
                                post_shift=post[1] if post else None, n_rows=n, n_out=w, k=K, activation=code, alpha=alpha,
                                out_hi_lo=None if last_hidden else out, out_f32=out if last_hidden else None,
                                tile=self._fused_tile(n, w, K), k_splits=1, range_flag=self.range_flag)
                    a = out
                    continue
                nxt = layers[i + 1]
                head_ok = last_hidden and self.fused_head and nxt[0] == "f32" and nxt[1].shape[0] <= 16 and w in (512, 1024) and res is None and not post
                if plan == "library":
                    part = torch.empty((2, n, w), dtype=torch.float32, device=a.device)
                    _mm_f32(a, B2.t(), part[0])          # hi x lo + lo x hi, scaled by 2^11
                    _mm_f32(a[:, :K], Wh.t(), part[1])   # hi x hi
                    n_corr = 1
                else:
                    # too few whole-K tiles to fill the chip: the own kernel with its K loop cut into chunks (one workgroup per tile and
                    # chunk, ~one per CU), raw fp32 partials; the first n_corr hold correction products only (still scaled by 2^11)
                    _, tile, chunks = plan
                    part = torch.empty((chunks, n, w), dtype=torch.float32, device=a.device)
                    _layer_call("rc_split_layer_f16", a=a, w=W3, n_rows=n, n_out=w, k=K, out_partials=part, k_splits=chunks, tile=tile)
                    n_corr = _hip.lib().rc_split_layer_corr_chunks(K, chunks)
                if head_ok:
                    # activation + the skinny output layer in one pass: the fp32 activations are never written (rc_head_split_f32)
                    out = torch.empty((n, 16), dtype=torch.float32, device=a.device)
                    if part.shape[0] == 2 and n_corr == 1:
                        c, c_corr, bias_h, act_h = part[1], part[0], b, code
                    else:   # more than two partials: summed (bias, activation) by the reduce kernel first
                        c, c_corr, bias_h, act_h = self._act(part, n_corr, b, code, alpha, split=False), None, self._zero_bias(w), 0
                    _hip.check(_hip.lib().rc_head_split_f32(c.data_ptr(), _ptr(c_corr), 1.0 / SPLIT_SCALE, n, w, bias_h.data_ptr(), act_h, alpha,
                                                            nxt[1].data_ptr(), nxt[2].data_ptr(), nxt[1].shape[0], out.data_ptr(),
                                                            _hip.stream_ptr()), "rc_head_split_f32")
                    return out[:, :nxt[1].shape[0]]
            else:
                _, W, b = layer
                return torch.addmm(b, a, W.t())
            a = self._act(part, n_corr, b, code, alpha, split=not last_hidden, skip=res, post=post)   # the output layer takes plain fp32 activations
        return a

    # ---- InferenceNet's interface --------------------------------------------------------------------------------
    @staticmethod
    def _cubes_from_oh(oh: torch.Tensor):
        """The states a one-hot batch encodes (cube.py:265-277 backwards), as device cubes: deterministic mode runs the input layer
        on the fused kernel for this entry point as well (the library GEMM on the one-hot matrix chooses its kernel by batch shape)."""
        from librubiks.cube.device import DeviceCubes
        codes = oh.reshape(oh.shape[0], 20, 24).argmax(2).to(torch.int8)
        return DeviceCubes.from_aos(codes.contiguous())

    @torch.no_grad()
    def __call__(self, oh: torch.Tensor):
        if self.deterministic:
            return self.forward_cubes(self._cubes_from_oh(oh))
        out = self._forward(self._input_from_oh(oh), self.layers)
        return out[:, :N_ACTIONS], out[:, N_ACTIONS]

    @torch.no_grad()
    def value(self, oh: torch.Tensor) -> torch.Tensor:
        if self.deterministic:
            return self.value_cubes(self._cubes_from_oh(oh))
        return self._forward(self._input_from_oh(oh), self.value_layers).reshape(-1)

    @torch.no_grad()
    def head_cubes(self, cubes, x1=None) -> torch.Tensor:
        """[n, 13] float32: 12 policy logits, then the value (the layout rc_mcts_backup_head reads)."""
        return self._forward_cubes(cubes, self.layers)

    @torch.no_grad()
    def forward_cubes(self, cubes, x1=None):
        out = self.head_cubes(cubes)
        return out[:, :N_ACTIONS], out[:, N_ACTIONS]

    @torch.no_grad()
    def value_cubes(self, cubes, x1=None, lo: int = 0, n: int = None) -> torch.Tensor:
        return self._forward_cubes(cubes, self.value_layers, lo, n).reshape(-1)


class _CubeWindow:
    """Rows lo .. lo + n of a DeviceCubes batch as (base pointer, n, stride) for the fused input layer."""
    __slots__ = ("ptr", "n", "stride")

    def __init__(self, ptr: int, n: int, stride: int):
        self.ptr, self.n, self.stride = ptr, n, stride


def _soa_ptr(cubes) -> int:
    return cubes.ptr if isinstance(cubes, _CubeWindow) else cubes.soa.data_ptr()


def _activate_(x: torch.Tensor, act: nn.Module) -> torch.Tensor:
    """In-place activation; bf16 tensors on the GPU go through rc_act_bf16_inplace (16-byte lanes), the rest through torch."""
    if x.dtype == torch.bfloat16 and x.is_cuda and x.is_contiguous() and x.numel() % 8 == 0:
        from librubiks import _hip
        relu = isinstance(act, nn.ReLU)
        _hip.check(_hip.lib().rc_act_bf16_inplace(x.data_ptr(), x.numel(), 1 if relu else 2, 0.0 if relu else float(act.alpha),
                                                  _hip.stream_ptr()), "rc_act_bf16_inplace")
        return x
    return F.relu_(x) if isinstance(act, nn.ReLU) else F.elu_(x, alpha=act.alpha)


class GenericNet:
    """Calls an arbitrary torch module with the reference's convention net(oh) -> [policy, value]."""

    input_dtype = torch.float32

    def __init__(self, module: nn.Module):
        self.module = module
        self.flops_per_state = None

    @torch.no_grad()
    def __call__(self, oh: torch.Tensor):
        p, v = self.module(oh)
        return p.float(), v.float().reshape(-1)

    @torch.no_grad()
    def value(self, oh: torch.Tensor) -> torch.Tensor:
        return self.module(oh, policy=False, value=True).float().reshape(-1)


def net_fingerprint(net, dtype=None):
    """
    Identity of the function an engine built from `net` would compute: the object, the requested dtype and, for a
    trainable `Model`, the storage and in-place version counter of every parameter and buffer (optimizer steps,
    `load_state_dict` and `.data` swaps all change it).  Engines (`InferenceNet`, `GenericNet`) are frozen or call
    their module live, so their identity is enough.  Search engines are rebuilt whenever this changes.
    """
    if isinstance(net, (InferenceNet, SplitF32Net, GenericNet)):
        return (id(net),)
    if isinstance(net, Model) and net.config.architecture.split("_")[0] in ("fc", "res"):
        tensors = list(net.parameters()) + list(net.buffers())
        return (id(net), str(dtype)) + tuple((t.data_ptr(), t._version) for t in tensors)
    return (id(net), str(dtype))   # any other module is called live through GenericNet


def make_inference_net(net, dtype=torch.bfloat16):
    """The fastest engine that preserves `net`'s eval-mode function."""
    if isinstance(net, (InferenceNet, SplitF32Net, GenericNet)):
        return net
    if isinstance(net, Model) and net.config.architecture.split("_")[0] in ("fc", "res"):
        if dtype in (F32_SPLIT, F32_SPLIT_DET):
            try:
                return SplitF32Net(net, deterministic=dtype == F32_SPLIT_DET)
            except SplitRangeError as e:   # the reference's fp32 forward has no such limit: run these weights on the fp32 GEMM chain
                if dtype == F32_SPLIT_DET:     # ... but not under a promise of bit-reproducibility, which that chain does not keep
                    raise
                warnings.warn(f"SplitF32Net: {e}; using the fp32 GEMM chain for this network", RuntimeWarning)
                return InferenceNet(net, dtype=torch.float32)
        return InferenceNet(net, dtype=dtype)
    return GenericNet(net)
