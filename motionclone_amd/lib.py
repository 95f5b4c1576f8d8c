"""ctypes binding of the C ABI in include/mc_kernels.h.

The product path opens exactly one library: csrc/libmotionclone_hip.so, built by hipcc for gfx950
(`python -m motionclone_amd.build` or `__graft_entry__.build()`).  If it is missing the import of any
op fails loudly - there is no eager / CPU fallback.

`use_library_for_tests()` lets the CPU-only test-suite point the same wrappers at the host simulator
build of the kernels (tests/hipemu); it is never called from the package itself.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MC_HIP_LIB: A/B tooling only (a build of the same sources with other compiler flags); the default is the in-tree library
HIP_LIB_PATH = os.environ.get("MC_HIP_LIB") or os.path.join(_HERE, "csrc", "libmotionclone_hip.so")

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p
P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> argument ctypes, in the order of include/mc_kernels.h
SIGNATURES = {
    "mc_version": [],
    "mc_workspace_bytes_gemm_splitk": [I, I, I],
    "mc_workspace_bytes_groupnorm": [I, I],
    "mc_workspace_bytes_groupnorm_bwd_stats": [I],
    "mc_workspace_bytes_attn_bwd": [I, I, I],
    "mc_workspace_bytes_tattn_loss": [I, I, I],
    "mc_gemm_f16": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, P],
    "mc_workspace_bytes_gemm_gnstats": [I, I],
    "mc_gemm_gnstats_f16": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, P, I, P],
    "mc_workspace_bytes_gemm_tileloop": [I],
    "mc_gemm_tileloop_f16": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, P, ctypes.c_size_t, P, ctypes.c_size_t, P],
    "mc_norm_gemm_f16": [P, P, P, P, I, I, I, I, I, I, P, P, P, I, I, F, P, P, I, P],
    "mc_gemm_splitk_plan": [I, I, I, I],
    "mc_gemm_last_kernel": [],
    "mc_attn_last_kernel": [],
    "mc_tattn_last_kernel": [],
    "mc_softmax_rows_f16": [P, I, I, I, P],
    "mc_video_post_f32": [P, I, P, I, I, I, P],
    "mc_vae_sample_f16": [P, I, P, P, I, I, I, P],
    "mc_video_resize_u8_f16": [P, P, I, I, I, I, I, I, P],
    "mc_attn_fwd_causal_f16": [P, P, P, I, I, I, P, I, I, I, I, I, F, P],
    "mc_clip_embed_f16": [P, P, P, P, I, I, I, I, P],
    "mc_quick_gelu_f16": [P, P, L, P],
    "mc_gemm_splitk_f16": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, F, I, P, I, P],
    "mc_gn_nchunk": [I],
    "mc_groupnorm_stats_f16": [P, P, I, I, I, I, I, I, F, P, P, P],
    "mc_groupnorm_apply_f16": [P, P, I, I, I, I, I, I, P, P, P, P, I, I, P],
    "mc_groupnorm_fwd_f16": [P, P, I, I, I, I, I, I, F, P, P, P, P, P, I, I, P],
    "mc_groupnorm_fwd_partial_f16": [P, I, I, I, I, F, P, I, P, P, P, P, I, I, P],
    "mc_groupnorm_bwd_f16": [P, P, I, I, I, I, I, I, P, I, P, P, P, I, P, P, P, I, I, P],
    "mc_layernorm_fwd_f16": [P, I, P, I, P, P, P, I, I, P, I, I, F, P],
    "mc_layernorm_bwd_f16": [P, I, P, I, P, P, P, I, P, I, I, I, P],
    "mc_attn_fwd_f16": [P, P, P, I, I, I, P, I, P, I, I, I, I, I, I, F, P],
    "mc_attn_bwd_f16": [P, P, P, I, I, I, P, I, P, I, P, P, P, I, P, I, P, I, I, I, I, I, I, I, F, P],
    "mc_tattn_fwd_f16": [P, P, P, I, P, I, I, I, I, I, I, F, P],
    "mc_tattn_top1_f16": [P, P, I, P, P, I, I, I, I, I, F, P],
    "mc_tattn_prob_f16": [P, P, I, P, I, I, I, I, I, F, P],
    "mc_tattn_loss_f16": [P, P, I, P, P, P, P, I, I, I, I, I, F, P],
    "mc_tattn_bwd_f16": [P, P, P, I, P, I, P, P, P, I, P, P, F, I, I, I, I, I, F, P],
    "mc_reduce_sum_f32": [P, L, F, P, P],
    "mc_geglu_fwd_f16": [P, I, P, I, I, I, P],
    "mc_geglu_bwd_f16": [P, I, P, I, P, I, I, I, P],
    "mc_add_f16": [P, I, P, I, P, I, I, I, F, F, P],
    "mc_sumpool2_f16": [P, I, P, I, I, I, I, I, I, P],
    "mc_latent_to_cl_f16": [P, P, I, I, I, I, I, P],
    "mc_cl_to_latent_f16": [P, I, P, I, F, I, I, I, I, P],
    "mc_timestep_embed_f16": [P, P, I, I, P],
    "mc_silu_f16": [P, P, L, P],
    "mc_cfg_ddim_step_f16": [P, P, I, P, P, P, P, F, F, F, F, F, F, I, I, I, P],
    "mc_ddim_step_general_f16": [P, P, P, P, P, P, P, L, F, F, F, F, F, I, F, F, F, F, F, F, P],
}

# exported by the TOOLS build only (-DMC_TOOLS: tools/_build/libmotionclone_hip_tools.so, the host simulator): bound when present
TOOLS_SIGNATURES = {
    "mc_gemm_debug": [I],
    "mc_gemm_debug_buffer": [P],
    "mc_tattn_debug_buffer": [P],
}

ERRORS = {-1: "bad shape / stride / alignment", -2: "unsupported size", -3: "kernel launch failed"}

_lib = None
_is_emulated = False


class KernelLibraryMissing(RuntimeError):
    pass


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = c_long if name.startswith("mc_workspace_bytes_") else c_int
    for name, argtypes in TOOLS_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = argtypes
            fn.restype = c_int
    return lib


def load():
    """Return the bound gfx950 library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise KernelLibraryMissing(
                "%s not found: build it with `python -m motionclone_amd.build` (hipcc, gfx950). "
                "There is no fallback path." % HIP_LIB_PATH)
        _lib = _bind(HIP_LIB_PATH)
    return _lib


def use_library_for_tests(path):
    """TEST HOOK: route the wrappers to the host-simulator build of the kernel sources."""
    global _lib, _is_emulated
    _lib = _bind(path)
    _is_emulated = True
    return _lib


def is_emulated():
    return _is_emulated


_FN = {}   # name -> bound ctypes function of the loaded library (cleared whenever the library changes)


_WS = {}   # (op, dims) -> workspace bytes, memoised (one ctypes round trip per distinct shape)


def workspace_bytes(op, *dims):
    """mc_workspace_bytes_<op>(dims...) of the loaded library"""
    key = (op, dims, _is_emulated)
    n = _WS.get(key)
    if n is None:
        n = getattr(load() if _lib is None else _lib, "mc_workspace_bytes_" + op)(*dims)
        if n < 0:
            raise RuntimeError("mc_workspace_bytes_%s%r: bad arguments" % (op, dims))
        _WS[key] = n
    return n


def try_call(name, *args):
    """like `call`, but MC_ERR_UNSUPPORTED (-2) is returned as False instead of raised: entry points that cover a subset of
    the shapes (mc_norm_gemm_f16) let the caller fall back to the general launch sequence"""
    if _lib is None or _FN.get("__lib__") is not _lib:
        _FN.clear()
        _FN["__lib__"] = load()
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib, name)
    rc = fn(*args)
    if rc == -2:
        return False
    if rc != 0:
        raise RuntimeError("%s failed: %s (rc=%d)" % (name, ERRORS.get(rc, "unknown"), rc))
    return True


def call_count(name, *args):
    """entry points that answer with a non-negative count on success (mc_gemm_gnstats_f16: the chunk height): the count, or
    False for MC_ERR_UNSUPPORTED (-2: nothing was launched, the caller issues the general sequence); other errors raise"""
    if _lib is None or _FN.get("__lib__") is not _lib:
        _FN.clear()
        _FN["__lib__"] = load()
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib, name)
    rc = fn(*args)
    if rc == -2:
        return False
    if rc < 0:
        raise RuntimeError("%s failed: %s (rc=%d)" % (name, ERRORS.get(rc, "unknown"), rc))
    return rc


def call(name, *args):
    if _lib is None or _FN.get("__lib__") is not _lib:
        _FN.clear()
        _FN["__lib__"] = load()
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(_lib, name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed: %s (rc=%d)" % (name, ERRORS.get(rc, "unknown"), rc))
