"""zk-franchise-proof-circuit_amd -- host-side mirror of the zkCensus proving interface over libzkcensus.so.

The reference's hot path sits behind snarkjs `groth16.fullProve / prove / verify` (ts_inputs/src/example.ts:358-362)
and dvote's `prover.Prove / ParseProof / Verify` (zk_census_test.go:89-122).  This package keeps those names and
argument meanings on top of the C ABI in include/zkcensus.h; all arithmetic runs in HIP kernels on the MI355X.
"""
import ctypes
from . import _native
from ._native import ZkcError
from .inputs import INPUT_KEYS, flatten_inputs, R_MOD

__all__ = ['Context', 'ProvingKey', 'DevicePool', 'ProvingService', 'groth16', 'ZkcError', 'INPUT_KEYS', 'flatten_inputs', 'R_MOD']


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
        self._keys = []            # ProvingKey handles living on this context: freed before the context itself

    def close(self):
        if getattr(self, '_h', None):
            for k in list(self._keys):
                k.close()
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


class ProvingKey:
    """A .zkey made device-resident (include/zkcensus.h zkc_zkey_load).  One proof in flight per handle."""

    def __init__(self, ctx, zkey_bytes):
        self.ctx = ctx
        self._lib = ctx._lib
        h = ctypes.c_void_p()
        ctx._check(self._lib.zkc_zkey_load(ctx._h, zkey_bytes, len(zkey_bytes), ctypes.byref(h)))
        self._h = h
        ctx._keys.append(self)
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._lib.zkc_zkey_info(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        self.n_vars, self.n_public, self.domain_size = a.value, b.value, c.value
        ps, ln = ctypes.c_int(), ctypes.c_int()
        self._lib.zkc_zkey_pass_info(h, ctypes.byref(ps), ctypes.byref(ln))
        self.pass_size, self.lanes = ps.value, ln.value          # a batch call of B voters runs as ceil(B / pass_size) equal passes over `lanes` pipeline lanes

    def close(self):
        if getattr(self, '_h', None):
            if getattr(self.ctx, '_h', None):          # the context frees its keys when it closes first
                self._lib.zkc_zkey_free(self._h)
            self._h = None
            if self in self.ctx._keys:
                self.ctx._keys.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _scalar(x):
        return x if isinstance(x, (bytes, bytearray)) else int(x).to_bytes(32, 'little')

    def prove(self, wtns, r, s):
        """wtns: n_vars x 32 B standard-form bytes (host).  Returns (proof 256 B, public n_public x 32 B)."""
        proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * self.n_public)
        self.ctx._check(self._lib.zkc_prove(self._h, wtns, len(wtns) // 32, self._scalar(r), self._scalar(s), proof, pub))
        return proof.raw, pub.raw

    def prove_dev(self, d_wtns_ptr, r, s):
        proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * self.n_public)
        self.ctx._check(self._lib.zkc_prove_dev(self._h, d_wtns_ptr, self.n_vars, self._scalar(r), self._scalar(s), proof, pub))
        return proof.raw, pub.raw

    def prove_batch_dev(self, d_wtns_ptr, B, rs):
        """rs: B x 64 bytes (r || s).  Returns (proofs B x 256 B, publics B x n_public x 32 B) as bytes."""
        proofs = ctypes.create_string_buffer(256 * B); pubs = ctypes.create_string_buffer(32 * self.n_public * B)
        self.ctx._check(self._lib.zkc_prove_batch_dev(self._h, d_wtns_ptr, self.n_vars, B, bytes(rs), proofs, pubs))
        return proofs.raw, pubs.raw

    def fullprove_batch_dev(self, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr, rs):
        """groth16.fullProve for a batch on the device: inputs (B x 334 x 32 B) -> witnesses (left in d_wtns), status (d_status) and proofs;
        witness generation of one pass overlaps the MSMs of the previous one.  Returns (proofs, publics) like prove_batch_dev."""
        proofs = ctypes.create_string_buffer(256 * B); pubs = ctypes.create_string_buffer(32 * self.n_public * B)
        self.ctx._check(self._lib.zkc_fullprove_batch_dev(self._h, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr, bytes(rs), proofs, pubs))
        return proofs.raw, pubs.raw

    def batch_begin(self, slot, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr, rs):
        """First half of a batch call (zkc_batch_begin): everything is enqueued, nothing is waited for.  d_inputs_ptr None: the witnesses are given.  The device
        buffers belong to the call until batch_finish(slot, B) returns; two slots may be in flight, so that one call's tail overlaps the next call's head."""
        self.ctx._check(self._lib.zkc_batch_begin(self._h, slot, d_inputs_ptr, B, d_wtns_ptr, d_status_ptr, bytes(rs)))

    def batch_finish(self, slot, B):
        proofs = ctypes.create_string_buffer(256 * B); pubs = ctypes.create_string_buffer(32 * self.n_public * B)
        self.ctx._check(self._lib.zkc_batch_finish(self._h, slot, proofs, pubs))
        return proofs.raw, pubs.raw

    def debug_stage(self, d_wtns_ptr, stage):
        out = ctypes.create_string_buffer((96 if stage == 0 else 32) * self.domain_size)
        self.ctx._check(self._lib.zkc_debug_stage(self._h, d_wtns_ptr, stage, out))
        return out.raw

    def msm_debug(self, which, d_scalars_ptr, count):
        out = ctypes.create_string_buffer(128 if which == 2 else 64)
        self.ctx._check(self._lib.zkc_msm_debug(self._h, which, d_scalars_ptr, count, out))
        return out.raw


class DevicePool:
    """Several GPUs from one host process (include/zkcensus.h zkc_pool_*): one context and one resident key per device, a batch split into
    contiguous blocks with one host thread per device.  What a single-process host of the reference (the Go loop over prover.Prove,
    zk_census_test.go:89) would hold; bench.py uses one process per GPU instead."""

    def __init__(self, devices, zkey_bytes=None):
        self._lib = _native.load()
        arr = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        rc = self._lib.zkc_pool_create(arr, len(devices), ctypes.byref(h))
        if rc != 0:
            raise ZkcError(rc, (self._lib.zkc_pool_last_error(None) or b'').decode())
        self._h = h
        self.devices = list(devices)
        self.n_public = None
        if zkey_bytes is not None:
            self.load_key(zkey_bytes)

    def _check(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise ZkcError(rc, (self._lib.zkc_pool_last_error(self._h) or b'').decode())
        return rc

    def load_key(self, zkey_bytes):
        self._check(self._lib.zkc_pool_zkey_load(self._h, zkey_bytes, len(zkey_bytes)))
        a, b, c = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
        self._lib.zkc_zkey_info(self._lib.zkc_pool_zkey(self._h, 0), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        self.n_vars, self.n_public, self.domain_size = a.value, b.value, c.value

    def fullprove_batch(self, inputs, rs=None, nLevels=160):
        """inputs: list of 12-key input objects or pre-flattened bytes.  rs: B x 64 bytes or None (drawn uniformly by the library).
        Returns (proofs B x 256 B, publics B x n_public x 32 B, status list); raises unless every voter passed or only circuit asserts failed."""
        flat = b''.join(x if isinstance(x, (bytes, bytearray)) else flatten_inputs(x, nLevels) for x in inputs)
        B = len(inputs)
        proofs = ctypes.create_string_buffer(256 * B); pubs = ctypes.create_string_buffer(32 * self.n_public * B); st = (ctypes.c_int32 * B)()
        self._check(self._lib.zkc_pool_fullprove_batch(self._h, flat, B, None if rs is None else bytes(rs), proofs, pubs, st), allow=(7,))
        return proofs.raw, pubs.raw, list(st)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.zkc_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ProvingService:
    """The submission queue behind the single-proof entry points (include/zkcensus.h zkc_service_*): callers -- one thread per voter, the way the
    reference's hosts call prover.Prove (zk_census_test.go:89) and groth16.fullProve (ts_inputs/src/example.ts:358-362) -- enqueue one voter each and
    a worker per GPU proves whatever has accumulated in one pipeline pass sequence.  The calls block; ctypes releases the GIL meanwhile."""

    def __init__(self, devices=None, default=False):
        self._lib = _native.load()
        self._own = not default
        if default:
            h = self._lib.zkc_service_default()
            if not h:
                raise ZkcError(6, (self._lib.zkc_service_last_error() or b'').decode())
            self._h = ctypes.c_void_p(h)
        else:
            devices = list(devices or [])
            arr = (ctypes.c_int * max(1, len(devices)))(*[int(d) for d in devices])
            h = ctypes.c_void_p()
            rc = self._lib.zkc_service_create(arr if devices else None, len(devices), ctypes.byref(h))
            if rc != 0:
                raise ZkcError(rc, (self._lib.zkc_service_last_error() or b'').decode())
            self._h = h

    def fullprove(self, zkey_bytes, inputs, nLevels=160, rs=None, n_public=8):
        """One voter: inputs = 12-key object or pre-flattened bytes.  Returns (proof 256 B, publics, status); raises on anything but a circuit assert."""
        flat = inputs if isinstance(inputs, (bytes, bytearray)) else flatten_inputs(inputs, nLevels)
        proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * n_public); st = ctypes.c_int32(0); err = ctypes.create_string_buffer(512)
        rc = self._lib.zkc_service_fullprove(self._h, zkey_bytes, len(zkey_bytes), nLevels, bytes(flat), None if rs is None else bytes(rs), proof, pub, ctypes.byref(st), err, 512)
        if rc not in (0, 7):
            raise ZkcError(rc, err.value.decode())
        return proof.raw, pub.raw, st.value

    def prove(self, zkey_bytes, wtns, rs=None, n_public=8):
        """One witness (n_vars x 32 B, standard form, host).  Returns (proof, publics)."""
        proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * n_public); err = ctypes.create_string_buffer(512)
        rc = self._lib.zkc_service_prove(self._h, zkey_bytes, len(zkey_bytes), wtns, len(wtns) // 32, None if rs is None else bytes(rs), proof, pub, err, 512)
        if rc != 0:
            raise ZkcError(rc, err.value.decode())
        return proof.raw, pub.raw

    def stats(self):
        out = (ctypes.c_uint64 * 8)()
        self._lib.zkc_service_stats(self._h, out)
        return dict(zip(('requests', 'batches', 'largest_batch', 'key_loads', 'devices', 'devices_used', 'failed', 'waiting'), [int(x) for x in out]))

    def timing(self):
        out = (ctypes.c_uint64 * 8)()
        self._lib.zkc_service_timing(self._h, out)
        return dict(zip(('us_upload', 'us_wait_gpu', 'us_key', 'us_prove', 'us_finish', 'proofs', 'batches', 'key_evictions'), [int(x) for x in out]))

    def memory(self):
        out = (ctypes.c_uint64 * 8)()
        self._lib.zkc_service_memory(self._h, out)
        return dict(zip(('resident_keys', 'table_bytes', 'work_bytes', 'largest_key_table_bytes', 'largest_device_work_bytes', 'staging_device_bytes', 'pinned_host_bytes', 'reserve_failures'),
                        [int(x) for x in out]))

    def close(self):
        if getattr(self, '_h', None) and self._own:
            self._lib.zkc_service_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from . import groth16  # noqa: E402  (snarkjs-shaped surface: groth16.fullProve / prove / verify, groth16.wtns.calculate)
