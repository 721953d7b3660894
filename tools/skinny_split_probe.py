"""Hidden layers of the split engine below the whole-K tile's fill point: the own kernel with its K loop cut into chunks
(SplitF32Net._k_split: ~256 workgroups, raw partials + rc_split_reduce_f16) against the two library GEMMs + reduce and against
the K loop cut in two (the round-2 'partials' plan).   python tools/skinny_split_probe.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "rl-rubiks_amd")]
from librubiks import _hip  # noqa: E402
from librubiks.model import SplitF32Net, _layer_call, _mm_f32  # noqa: E402

lib = _hip.lib()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for K, N in ((4096, 2048), (2048, 1024)):
    W3 = torch.randn(N, 3 * K, device="cuda").half()
    Wh, B2 = W3[:, 2 * K:].contiguous(), W3[:, :2 * K].contiguous()
    bias = torch.randn(N, device="cuda")
    for M in (352, 704, 1056, 1408, 2112, 2816, 3520, 4224, 5632, 7040, 8448, 9856, 11264):
        a = (torch.randn(M, 2 * K, device="cuda") * 0.5).half()
        out = torch.empty((M, 2 * N), dtype=torch.float16, device="cuda")
        part2 = torch.empty((2, M, N), device="cuda")

        def reduce(part, S, n_corr):
            _hip.check(lib.rc_split_reduce_f16(part.data_ptr(), M * N, S, n_corr, M, N, bias.data_ptr(), None, 2, 1.0, None, None, out.data_ptr(), None,
                                               None, _hip.stream_ptr()))

        def library():
            _mm_f32(a, B2.t(), part2[0])
            _mm_f32(a[:, :K], Wh.t(), part2[1])
            reduce(part2, 2, 1)

        def halves():
            _layer_call("rc_split_layer_f16", a=a, w=W3, n_rows=M, n_out=N, k=K, out_partials=part2, k_splits=2)
            reduce(part2, 2, 1)

        def whole():
            _layer_call("rc_split_layer_f16", a=a, w=W3, bias=bias, n_rows=M, n_out=N, k=K, activation=2, alpha=1.0, out_hi_lo=out, tile=1, k_splits=1)
        tiles = -(-M // 352) * (N // 256)
        line = f"K={K} N={N} M={M:5d} ({tiles:3d} tiles): library {t(library):6.1f} us  K/2 {t(halves):6.1f}  whole {t(whole):6.1f}"
        cut = SplitF32Net._k_split(M, N, K)
        if cut:
            tile, S = cut
            part = torch.empty((S, M, N), device="cuda")
            n_corr = lib.rc_split_layer_corr_chunks(K, S)

            def own():
                _layer_call("rc_split_layer_f16", a=a, w=W3, n_rows=M, n_out=N, k=K, out_partials=part, k_splits=S, tile=tile)
                reduce(part, S, n_corr)
            line += f"  cut {'128c' if tile == 3 else '256c'} x{S:2d} {t(own):6.1f}"
        print(line, flush=True)
