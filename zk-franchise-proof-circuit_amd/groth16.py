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


def vk_to_bytes(vk):
    """verification_key.json object -> the binary layout of zkc_verify_bin / zkc_verify_batch (standard form, little endian)."""
    le = lambda d: int(d).to_bytes(32, 'little')
    g1 = lambda p: le(p[0]) + le(p[1])
    g2 = lambda p: le(p[0][0]) + le(p[0][1]) + le(p[1][0]) + le(p[1][1])
    return g1(vk['vk_alpha_1']) + g2(vk['vk_beta_2']) + g2(vk['vk_gamma_2']) + g2(vk['vk_delta_2']) + b''.join(g1(p) for p in vk['IC'])


def verify_batch(ctx, vk, publics, proofs, seed=None):
    """Batch counterpart of verify (no snarkjs equivalent): N proofs under one key in one random-linear-combination pairing check.
    vk: parsed verification_key.json (or its binary form); publics: N x nPublic x 32 bytes, proofs: N x 256 bytes, as returned by
    ProvingKey.prove_batch_dev.  seed: 32 bytes of fresh randomness (None: OS).  True iff every proof is valid."""
    lib = _native.load()
    vkb = vk if isinstance(vk, (bytes, bytearray)) else vk_to_bytes(vk)
    n = len(proofs) // 256
    npub = (len(vkb) - 448) // 64 - 1
    if n == 0 or len(proofs) != 256 * n or len(publics) != 32 * npub * n:
        raise ValueError('verify_batch: proofs must be N x 256 bytes and publics N x nPublic x 32 bytes')
    rc = lib.zkc_verify_batch(ctx._h, bytes(vkb), npub, bytes(publics), bytes(proofs), n, seed)
    if rc < 0:
        raise _native.ZkcError(-rc, (lib.zkc_verify_last_error() or b'').decode())
    return rc == 1
