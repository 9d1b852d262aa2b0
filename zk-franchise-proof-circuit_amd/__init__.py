"""zk-franchise-proof-circuit_amd -- host-side mirror of the zkCensus proving interface over libzkcensus.so.

The reference's hot path sits behind snarkjs `groth16.fullProve / prove / verify` (ts_inputs/src/example.ts:358-362)
and dvote's `prover.Prove / ParseProof / Verify` (zk_census_test.go:89-122).  This package keeps those names and
argument meanings on top of the C ABI in include/zkcensus.h; all arithmetic runs in HIP kernels on the MI355X.
"""
import ctypes
from . import _native
from ._native import ZkcError
from .inputs import INPUT_KEYS, flatten_inputs, R_MOD

__all__ = ['Context', 'ZkcError', 'INPUT_KEYS', 'flatten_inputs', 'R_MOD']


class Context:
    """One prover context per (process, GPU): owns a HIP stream, the Poseidon tables and witness templates."""

    def __init__(self, device=0):
        self._lib = _native.load()
        h = ctypes.c_void_p()
        rc = self._lib.zkc_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise ZkcError(rc, (self._lib.zkc_last_error(None) or b'').decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, '_h', None):
            self._lib.zkc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise ZkcError(rc, (self._lib.zkc_last_error(self._h) or b'').decode())
        return rc

    @property
    def stream(self):
        return self._lib.zkc_ctx_stream(self._h)

    def n_wires(self, nLevels=160):
        return self._lib.zkc_circuit_n_wires(nLevels)

    def n_inputs(self, nLevels=160):
        return self._lib.zkc_circuit_n_inputs(nLevels)

    def witness(self, inputs, nLevels=160):
        """inputs: list of 12-key input objects (or pre-flattened bytes).  Returns (list of wtns bytes, list of status)."""
        flat = b''.join(x if isinstance(x, (bytes, bytearray)) else flatten_inputs(x, nLevels) for x in inputs)
        B = len(inputs)
        nw = self.n_wires(nLevels)
        out = ctypes.create_string_buffer(B * nw * 32)
        st = (ctypes.c_int32 * B)()
        self._check(self._lib.zkc_witness(self._h, nLevels, flat, B, out, st), allow=(7,))
        raw = out.raw
        return [raw[i * nw * 32:(i + 1) * nw * 32] for i in range(B)], list(st)

    def witness_dev(self, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr, nLevels=160):
        self._check(self._lib.zkc_witness_dev(self._h, nLevels, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr))
