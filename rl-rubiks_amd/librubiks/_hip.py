"""
ctypes binding of librubiks_hip.so (include/rubiks_hip.h).  The library is built in-tree by
`make -C rl-rubiks_amd` (or __graft_entry__.build()) into rl-rubiks_amd/lib/.

Fails loudly: a missing library, a missing symbol or a missing GPU raise -- nothing falls back to
the CPU.
"""
import ctypes
import os
from ctypes import c_char_p, c_int, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RUBIKS_HIP_LIB: another build of the same library (same-box A/B of kernel changes); it must export the same ABI
LIB_PATH = os.environ.get("RUBIKS_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "librubiks_hip.so")

P, SZ, I = c_void_p, c_size_t, c_int

# name -> argtypes (restype is always int unless listed in _RESTYPES); mirrors include/rubiks_hip.h
SIGNATURES = {
    "rc_abi_version": [],
    "rc_error_string": [I],
    "rc_init": [I],
    "rc_get_move_table": [P],
    "rc_get_solved": [P],
    "rc_aos_to_soa": [P, P, SZ, SZ, P],
    "rc_soa_to_aos": [P, P, SZ, SZ, P],
    "rc_multi_rotate_aos": [P, P, P, SZ, P],
    "rc_is_solved_aos": [P, P, SZ, P],
    "rc_as_oh_aos_f32": [P, P, SZ, P],
    "rc_multi_rotate": [P, P, P, SZ, SZ, SZ, P],
    "rc_expand12": [P, P, SZ, SZ, SZ, P],
    "rc_expand12_flags": [P, P, SZ, SZ, SZ, P, P, P],
    "rc_is_solved": [P, P, P, P, SZ, SZ, P],
    "rc_as_oh_f32": [P, P, SZ, SZ, P],
    "rc_as_oh_bf16": [P, P, SZ, SZ, P],
    "rc_apply_moves": [P, P, SZ, SZ, SZ, P],
    "rc_sequence_states": [P, P, SZ, SZ, I, SZ, P],
    "rc_act_bf16_inplace": [P, SZ, I, ctypes.c_float, P],
    "rc_oh_split_f16": [P, SZ, SZ, P, P],
    "rc_first_layer_split_f16": [P, SZ, SZ, P, P, P, P, SZ, I, ctypes.c_float, P],
    "rc_first_layer_split_flag_f16": [P, SZ, SZ, P, P, P, P, SZ, I, ctypes.c_float, P, P],
    "rc_first_layer_gather_f16": [P, SZ, SZ, P, P, P, SZ, I, ctypes.c_float, P, P],
    "rc_split_act_f16": [P, P, ctypes.c_float, SZ, SZ, P, I, ctypes.c_float, P, P, P],
    "rc_split_reduce_f16": [P, SZ, I, I, SZ, SZ, P, P, I, ctypes.c_float, P, P, P, P, P, P],
    "rc_split_layer_f16": [P, P],
    "rc_gemm_layer_bf16": [P, P],
    "rc_split_layer_corr_chunks": [SZ, I],
    "rc_split_layer_struct_bytes": [],
    "rc_split_gemm_f16": [P, P, P, SZ, SZ, SZ, I, ctypes.c_float, P, P, I, P],
    "rc_gemm_bias_act_bf16": [P, P, P, SZ, SZ, SZ, I, ctypes.c_float, P, I, P],
    "rc_split_gemm_partials_f16": [P, P, SZ, SZ, SZ, P, P],
    "rc_head_split_f32": [P, P, ctypes.c_float, SZ, SZ, P, I, ctypes.c_float, P, P, SZ, P, P],
    "rc_head_bf16": [P, SZ, SZ, P, P, SZ, P, I, ctypes.c_float, P],
    "rc_adi_targets": [P, P, P, SZ, SZ, ctypes.c_float, I, P, P, P],
    "rc_first_layer_mfma_bf16": [P, SZ, SZ, P, P, P, SZ, I, ctypes.c_float, I, P],
}
_RESTYPES = {"rc_error_string": c_char_p, "rc_split_layer_struct_bytes": c_size_t}

_lib = None
_initialised_devices = set()


class RubiksHipError(RuntimeError):
    pass


def register(signatures: dict, restypes: dict = None):
    """Lets the search modules declare their own entry points next to where they are used."""
    SIGNATURES.update(signatures)
    if restypes:
        _RESTYPES.update(restypes)
    if _lib is not None:
        _bind(_lib, signatures)


def _bind(lib, signatures):
    for name, argtypes in signatures.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RubiksHipError(f"librubiks_hip.so does not export {name}; rebuild it (make -C rl-rubiks_amd)") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)


def load() -> ctypes.CDLL:
    """Loads the shared library (no GPU needed for that) and binds every declared symbol."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RubiksHipError(f"{LIB_PATH} is missing: build it with `make -C rl-rubiks_amd` "
                                 "(there is no CPU fallback for the cube environment)")
        lib = ctypes.CDLL(LIB_PATH)
        _bind(lib, SIGNATURES)
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().rc_error_string(rc).decode()
        raise RubiksHipError(f"{what or 'librubiks_hip'} failed ({rc}): {msg}")


def lib() -> ctypes.CDLL:
    """The library, initialised for the current torch CUDA(HIP) device.  Raises without a GPU."""
    l = load()
    if not torch.cuda.is_available():
        raise RubiksHipError("no MI355X visible (torch.cuda.is_available() is False): the cube "
                             "environment has no CPU path")
    dev = torch.cuda.current_device()
    if dev not in _initialised_devices:
        check(l.rc_init(dev), "rc_init")
        _initialised_devices.add(dev)
    return l


def stream_ptr() -> int:
    """hipStream_t of torch's current stream, so kernels order with torch ops and capture into graphs."""
    return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor) -> int:
    return t.data_ptr()
