"""snarkjs-shaped surface over libzkcensus: groth16.fullProve / prove / verify and wtns.calculate
(ts_inputs/src/example.ts:1,358-362 imports `groth16` from snarkjs and calls `groth16.fullProve(inputs, wasm, zkey)`).

Same argument meaning and result shapes as snarkjs 0.7.0: proof = {pi_a, pi_b, pi_c, protocol, curve} of decimal strings,
publicSignals = list of decimal strings.  The `wasm_file` argument names the circuit the way it does for snarkjs: its SHA-256 selects
the native witness generator (80a73567...c139 = the reference's dev/160 circuit.wasm, artifacts/zkCensus/dev/circuits-info.md:7); a
wasm this build has no native circuit for is refused loudly here -- Python has no wasm runtime; the Node surface (napi/) executes such a wasm the way snarkjs does and
proves from its witness on the GPU.  `wasm_file=None` selects the native ZkFranchiseProofCircuit(nLevels) directly: the explicit `nLevels`, or the depth read off the key.  An optional
`rs=(r, s)` makes the proof deterministic for parity tests (snarkjs draws them at random)."""
import collections
import ctypes
import hashlib
import json
import os
import secrets

from . import _native
from .inputs import R_MOD

_ctx = None
_keys = collections.OrderedDict()      # sha256 of the .zkey image -> ProvingKey; at most MAX_RESIDENT_KEYS stay in HBM
_path_digest = {}                       # (path, mtime_ns, size) -> sha256, so that an unchanged file is not re-read per proof
def _max_resident_keys():
    """As many as the proving service keeps per device ($ZKC_SERVICE_KEYS), clamped the way the C side clamps it (1 .. 64; anything that is not a number: 4)."""
    try:
        return max(1, min(int(os.environ.get("ZKC_SERVICE_KEYS", "4")), 64))
    except ValueError:
        return 4


MAX_RESIDENT_KEYS = _max_resident_keys()


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


def _fingerprint(raw):
    out = ctypes.create_string_buffer(32)
    if _native.load().zkc_zkey_fingerprint(raw, len(raw), out) != 0:
        return hashlib.sha256(raw).digest()            # not a zkey: let the loader report it
    return out.raw


def _key(zkey_file):
    """Resident proving key for a .zkey given as path / bytes / {type: 'mem'}.  Identity = zkc_zkey_fingerprint of the image (SHA-256 over the
    header, IC, section ends and sampled blocks: cheap enough per call): a rewritten file or a second bytes object never gets a stale key."""
    from . import ProvingKey
    raw = None
    if isinstance(zkey_file, str):
        st = os.stat(zkey_file); ident = (os.path.abspath(zkey_file), st.st_mtime_ns, st.st_size)
        digest = _path_digest.get(ident)
        if digest is None:
            raw = _read(zkey_file); digest = _fingerprint(raw)
            _path_digest.clear(); _path_digest[ident] = digest
    else:
        raw = _read(zkey_file); digest = _fingerprint(raw)
    if digest in _keys:
        _keys.move_to_end(digest)
        return _keys[digest]
    while len(_keys) >= MAX_RESIDENT_KEYS:
        _, old = _keys.popitem(last=False)
        old.close()
    _keys[digest] = ProvingKey(_context(), raw if raw is not None else _read(zkey_file))
    return _keys[digest]


def circuit_nlevels(wasm_file, nLevels=None):
    """nLevels of the native circuit a snarkjs-style call names.  wasm_file None: the explicit nLevels (default 160)."""
    if wasm_file is None:
        return 160 if nLevels is None else int(nLevels)
    raw = _read(wasm_file)
    hexbuf = ctypes.create_string_buffer(65)
    nl = _native.load().zkc_circuit_nlevels_from_wasm(raw, len(raw), hexbuf)
    if nl < 0:
        raise ValueError('unknown circuit wasm (sha256 %s): this build has a native witness generator for the zkCensus circuit only '
                         '(dev/160 circuit.wasm, sha256 80a73567...c139) and does not execute wasm' % hexbuf.value.decode())
    if nLevels is not None and int(nLevels) != nl:
        raise ValueError('wasm file is the nLevels=%d circuit but nLevels=%s was requested' % (nl, nLevels))
    return nl


def proof_to_json(proof, pub):
    lib = _native.load()
    n = len(pub) // 32
    ps, us = ctypes.c_ulong(4096), ctypes.c_ulong(128 * n + 16)
    pb, ub = ctypes.create_string_buffer(ps.value), ctypes.create_string_buffer(us.value)
    rc = lib.zkc_proof_to_json(proof, pub, n, pb, ctypes.byref(ps), ub, ctypes.byref(us))
    if rc != 0:
        raise _native.ZkcError(rc, 'zkc_proof_to_json')
    return json.loads(pb.value.decode()), json.loads(ub.value.decode())


def status_text(nLevels, status):
    """The Error.message snarkjs carries for a per-voter witness status ("Assert Failed.\\nError in template ... line: N\\n"): zkc_witness_status_text"""
    t = _native.load().zkc_witness_status_text(int(nLevels), int(status))
    return t.decode() if t else 'Assert Failed.\n(witness status %d)' % status


class wtns:
    @staticmethod
    def calculate(inputs, wasm_file=None, nLevels=None):
        """Returns the .wtns file image (bytes).  Raises like snarkjs when a circuit assert fails."""
        nLevels = circuit_nlevels(wasm_file, nLevels)
        ctx = _context()
        ws, st = ctx.witness([inputs], nLevels)
        if st[0] != 0:
            raise RuntimeError(status_text(nLevels, st[0]))
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


def key_nlevels(zkey_file):
    """The depth n for which the key has the shape of ZkFranchiseProofCircuit(n) -- 8 public signals, zkc_circuit_n_wires(n) wires -- read off the file header alone
    (zkc_zkey_header_info, host only); None for any other circuit."""
    lib = _native.load()
    raw = _read(zkey_file)
    nv, npub, dom = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint32()
    if lib.zkc_zkey_header_info(raw, len(raw), ctypes.byref(nv), ctypes.byref(npub), ctypes.byref(dom)) != 0:
        raise ValueError('not a Groth16 .zkey file')
    if npub.value != 8:
        return None
    return next((n for n in range(3, 254) if lib.zkc_circuit_n_wires(n) == nv.value), None)


def fullProve(inputs, wasm_file, zkey_file, rs=None, nLevels=None):
    if wasm_file is None and nLevels is None:           # [r4] no wasm names the circuit: the depth is the key's (a caller with a key of another depth needs no option)
        nLevels = key_nlevels(zkey_file)
        if nLevels is None:
            raise ValueError('no wasm_file given and the key is not a ZkFranchiseProofCircuit key: compute the witness elsewhere and call groth16.prove(zkey_file, wtns_file)')
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
    """verification_key.json (object or text) -> the binary layout of zkc_verify_bin / zkc_verify_batch (standard form, little endian), read by the library's
    strict parser (zkc_vkey_from_json)."""
    lib = _native.load()
    text = vk.encode() if isinstance(vk, str) else bytes(vk) if isinstance(vk, (bytes, bytearray)) else json.dumps(vk).encode()
    size = ctypes.c_ulong(0)
    rc = lib.zkc_vkey_from_json(text, None, ctypes.byref(size), None)
    if rc == -2:                                         # ZKC_ERR_SHORT_BUFFER: the size was written back
        buf = ctypes.create_string_buffer(size.value)
        rc = lib.zkc_vkey_from_json(text, buf, ctypes.byref(size), None)
    if rc != 1:
        raise _native.ZkcError(-rc, (lib.zkc_verify_last_error() or b'').decode())
    return buf.raw[:size.value]


def proof_from_json(proof, public_signals):
    """prover.ParseProof (zk_census_test.go:118): proof.json / signals.json (objects or texts) -> (256-byte proof, nPublic x 32-byte signals), the forms
    verify_batch and the binary entry points take.  Raises on a document json.Unmarshal would refuse; ValueError on a value that is no encoding."""
    lib = _native.load()
    t = lambda x: x.encode() if isinstance(x, str) else bytes(x) if isinstance(x, (bytes, bytearray)) else json.dumps(x).encode()
    n = ctypes.c_int(4096); pr = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * 4096)
    rc = lib.zkc_proof_from_json(t(proof), t(public_signals), pr, pub, ctypes.byref(n))
    if rc < 0:
        raise _native.ZkcError(-rc, (lib.zkc_verify_last_error() or b'').decode())
    if rc == 0:
        raise ValueError('proof / public signals hold a value that is no field or point encoding')
    return pr.raw, pub.raw[:32 * n.value]


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
