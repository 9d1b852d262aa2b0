"""snarkjs-shaped surface over libzkcensus: groth16.fullProve / prove / verify and wtns.calculate
(ts_inputs/src/example.ts:1,358-362 imports `groth16` from snarkjs and calls `groth16.fullProve(inputs, wasm, zkey)`).

Same argument meaning and result shapes as snarkjs 0.7.0: proof = {pi_a, pi_b, pi_c, protocol, curve} of decimal strings,
publicSignals = list of decimal strings.  The `wasm_file` argument is accepted for drop-in compatibility and only used to
select the circuit shape (its witness calculator is replaced by the HIP kernels); an optional `rs=(r, s)` makes the proof
deterministic for parity tests (snarkjs draws them at random)."""
import ctypes
import json
import os
import secrets

from . import _native
from .inputs import R_MOD

_ctx = None
_keys = {}


def _context(device=None):
    global _ctx
    if _ctx is None:
        from . import Context
        _ctx = Context(int(os.environ.get('ZKC_DEVICE', '0')) if device is None else device)
    return _ctx


def _read(f):
    if isinstance(f, (bytes, bytearray)):
        return bytes(f)
    if isinstance(f, dict) and f.get('type') == 'mem':
        return bytes(f['data'])
    with open(f, 'rb') as fh:
        return fh.read()


def _key(zkey_file):
    from . import ProvingKey
    ident = zkey_file if isinstance(zkey_file, str) else id(zkey_file)
    if ident not in _keys:
        _keys[ident] = ProvingKey(_context(), _read(zkey_file))
    return _keys[ident]


def proof_to_json(proof, pub):
    lib = _native.load()
    n = len(pub) // 32
    ps, us = ctypes.c_ulong(4096), ctypes.c_ulong(128 * n + 16)
    pb, ub = ctypes.create_string_buffer(ps.value), ctypes.create_string_buffer(us.value)
    rc = lib.zkc_proof_to_json(proof, pub, n, pb, ctypes.byref(ps), ub, ctypes.byref(us))
    if rc != 0:
        raise _native.ZkcError(rc, 'zkc_proof_to_json')
    return json.loads(pb.value.decode()), json.loads(ub.value.decode())


class wtns:
    @staticmethod
    def calculate(inputs, wasm_file=None, nLevels=160):
        """Returns the .wtns file image (bytes).  Raises like snarkjs when a circuit assert fails."""
        ctx = _context()
        ws, st = ctx.witness([inputs], nLevels)
        if st[0] != 0:
            sites = {1: 'ZkFranchiseProofCircuit line: 72', 2: 'ZkFranchiseProofCircuit line: 90', 3: 'ZkFranchiseProofCircuit line: 103',
                     4: 'ZkFranchiseProofCircuit line: 114', 5: 'SMTLevIns line: 93', 6: 'input >= field order'}
            raise RuntimeError('Error: Assert Failed. Error in template ' + sites.get(st[0], str(st[0])))
        lib = _native.load()
        n = len(ws[0]) // 32
        need = lib.zkc_wtns_write(ws[0], n, None, 0)
        out = ctypes.create_string_buffer(need)
        lib.zkc_wtns_write(ws[0], n, out, need)
        return out.raw


def _wtns_payload(wtns_file):
    lib = _native.load()
    raw = _read(wtns_file)
    ptr, n = ctypes.c_void_p(), ctypes.c_uint32()
    if lib.zkc_wtns_parse(raw, len(raw), ctypes.byref(ptr), ctypes.byref(n)) != 0:
        raise ValueError('Invalid witness file')
    off = ptr.value - ctypes.cast(ctypes.c_char_p(raw), ctypes.c_void_p).value
    return raw[off:off + 32 * n.value]


def prove(zkey_file, wtns_file, rs=None):
    pk = _key(zkey_file)
    r, s = rs if rs is not None else (secrets.randbelow(R_MOD), secrets.randbelow(R_MOD))
    proof, pub = pk.prove(_wtns_payload(wtns_file), r, s)
    pj, sj = proof_to_json(proof, pub)
    return {'proof': pj, 'publicSignals': sj}


def fullProve(inputs, wasm_file, zkey_file, rs=None, nLevels=160):
    return prove(zkey_file, wtns.calculate(inputs, wasm_file, nLevels), rs)


def verify(vk, public_signals, proof):
    """vk / proof: parsed JSON objects (or JSON text); public_signals: list of decimal strings.  Runs on the CPU."""
    lib = _native.load()
    t = lambda x: x.encode() if isinstance(x, str) else json.dumps(x).encode()
    rc = lib.zkc_verify(t(vk), t(public_signals), t(proof))
    if rc < 0:
        raise _native.ZkcError(-rc, (lib.zkc_verify_last_error() or b'').decode())
    return rc == 1
