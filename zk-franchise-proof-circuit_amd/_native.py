"""ctypes loader for libzkcensus.so (the C ABI declared in include/zkcensus.h).

The product has no CPU path: if the HIP library is missing or no GPU is visible, everything here raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ZKCENSUS_LIB') or os.path.join(_HERE, 'libzkcensus.so')      # ZKCENSUS_LIB: the same override the N-API addon honours
_lib = None


class ZkcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('libzkcensus error %d: %s' % (code, msg))
        self.code = code


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('libzkcensus.so is not built (run `python -c "import __graft_entry__ as g; g.build()"`); '
                          'there is no CPU fallback for the product path')
    try:
        # torch ships its own libamdhip64.so.7; both libraries resolve that soname, and torch only finds its GPUs when ITS copy
        # is the one loaded first.  Outside Python (N-API / cgo hosts) the system ROCm runtime is used.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)
    L.zkc_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.zkc_ctx_destroy.argtypes = [vp]
    L.zkc_last_error.argtypes = [vp]; L.zkc_last_error.restype = ctypes.c_char_p
    L.zkc_ctx_stream.argtypes = [vp]; L.zkc_ctx_stream.restype = vp
    L.zkc_witness.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, i32p]
    L.zkc_witness_dev.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp]
    u8p, u32p = ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)
    L.zkc_zkey_load.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.zkc_zkey_free.argtypes = [vp]; L.zkc_zkey_free.restype = None
    L.zkc_zkey_info.argtypes = [vp, u32p, u32p, u32p]
    L.zkc_zkey_pass_info.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.zkc_witness_status_text.argtypes = [ctypes.c_int, ctypes.c_int32]; L.zkc_witness_status_text.restype = ctypes.c_char_p
    L.zkc_prove.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint32, u8p, u8p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_prove_dev.argtypes = [vp, vp, ctypes.c_uint32, u8p, u8p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_prove_batch_dev.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_fullprove_batch_dev.argtypes = [vp, vp, ctypes.c_int, vp, vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_debug_stage.argtypes = [vp, vp, ctypes.c_int, ctypes.c_char_p]
    L.zkc_debug_early_retries.argtypes = []; L.zkc_debug_early_retries.restype = ctypes.c_ulonglong
    L.zkc_msm_debug.argtypes = [vp, ctypes.c_int, vp, ctypes.c_uint32, ctypes.c_char_p]
    ulp = ctypes.POINTER(ctypes.c_ulong)
    L.groth16_prover.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.c_char_p, ctypes.c_ulong, ctypes.c_char_p, ulp, ctypes.c_char_p, ulp, ctypes.c_char_p, ctypes.c_ulong]
    L.zkc_verify.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_verify_bin.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_verify_batch.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p]
    L.zkc_verify_last_error.restype = ctypes.c_char_p
    L.zkc_proof_to_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ulp, ctypes.c_char_p, ulp]
    L.zkc_proof_from_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
    L.zkc_vkey_from_json.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ulp, ctypes.POINTER(ctypes.c_int)]
    L.zkc_wtns_parse.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint32)]
    L.zkc_wtns_write.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_ulong]; L.zkc_wtns_write.restype = ctypes.c_ulong
    L.zkc_poseidon_batch.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    L.zkc_smt_build.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, i32p]
    L.zkc_census_inputs.argtypes = [vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, vp, vp, ctypes.c_char_p]
    L.zkc_profile_enable.argtypes = [vp, ctypes.c_uint32]
    L.zkc_ntt_dev.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.zkc_g1_mul_batch_dev.argtypes = [vp, ctypes.c_char_p, vp, ctypes.c_uint32, vp]
    L.zkc_msm_g1_load_dev.argtypes = [vp, vp, ctypes.c_uint32, ctypes.POINTER(vp)]
    L.zkc_msm_g1_dev.argtypes = [vp, vp, ctypes.c_char_p]
    L.zkc_msm_g1_free.argtypes = [vp]; L.zkc_msm_g1_free.restype = None
    L.zkc_profile_read.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.zkc_setup_from_r1cs.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkc_circuit_nlevels_from_wasm.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    L.zkc_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]; L.zkc_sha256.restype = None
    L.zkc_zkey_sha256.argtypes = [vp, ctypes.c_char_p]
    L.zkc_zkey_fingerprint.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    L.zkc_random_scalars.argtypes = [ctypes.c_char_p, ctypes.c_size_t]; L.zkc_random_scalars.restype = None
    L.zkc_pairing_bin.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_pool_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.zkc_pool_destroy.argtypes = [vp]; L.zkc_pool_destroy.restype = None
    L.zkc_pool_size.argtypes = [vp]
    L.zkc_pool_ctx.argtypes = [vp, ctypes.c_int]; L.zkc_pool_ctx.restype = vp
    L.zkc_pool_zkey.argtypes = [vp, ctypes.c_int]; L.zkc_pool_zkey.restype = vp
    L.zkc_pool_last_error.argtypes = [vp]; L.zkc_pool_last_error.restype = ctypes.c_char_p
    L.zkc_pool_zkey_load.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.zkc_pool_fullprove_batch.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, i32p]
    L.zkc_batch_begin.argtypes = [vp, ctypes.c_int, vp, ctypes.c_int, vp, vp, ctypes.c_char_p]
    L.zkc_batch_finish.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p]
    L.zkc_service_create.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)]
    L.zkc_service_destroy.argtypes = [vp]; L.zkc_service_destroy.restype = None
    L.zkc_service_default.argtypes = []; L.zkc_service_default.restype = vp
    L.zkc_service_last_error.restype = ctypes.c_char_p
    L.zkc_service_fullprove.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, i32p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkc_service_prove.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkc_service_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.zkc_service_timing.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.zkc_inputs_from_json.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.zkc_service_fullprove_json.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, i32p, ctypes.c_char_p, ctypes.c_size_t]
    L.groth16_fullprove.argtypes = [ctypes.c_char_p, ctypes.c_ulong, ctypes.c_char_p, ctypes.c_ulong, ctypes.c_char_p, ctypes.c_ulong, ctypes.c_char_p, ulp, ctypes.c_char_p, ulp, ctypes.c_char_p, ctypes.c_ulong]
    L.zkc_service_memory.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.zkc_service_submit_fullprove.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, vp, vp]
    _lib = L
    return L


def declared_symbols():
    """Every `zkc_*` / `groth16_*` function name declared in include/zkcensus.h (used by the CPU export test)."""
    import re
    hdr = open(os.path.join(_HERE, '..', 'include', 'zkcensus.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b((?:zkc|groth16)_[a-z0-9_]+)\s*\(', hdr)))
