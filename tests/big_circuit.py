"""TEST INFRASTRUCTURE: random satisfiable R1CS instances of 10^5 .. 10^6 constraints, written straight to an iden3 .r1cs file.

BASELINE configs[4] asks for a "~2^20-constraint R1CS"; the ceiling comes from the reference's own ceremony (`snarkjs powersoftau new bn128 20`,
circuit/circuit-compiler.sh:57).  No circuit of that size exists in the reference, so the instance is synthetic -- as tests/test_generic_circuit.py's small ones are, but
generated without per-constraint modular inversions or dict objects so that 2^20 rows take tens of seconds of Python, not minutes:

    row k :  <A_k, w> . <B_k, w> = c_k . w[i_k]        A_k: 1-3 terms, B_k: 1-2 terms, every 7th A_k a constant on wire 0; c_k = a_k b_k / w[i_k]

with 1 / w[i] from one batch inversion.  A tenth of the wires are booleans (0 / 1) -- real witnesses are full of them and they make zero digits and crowded buckets in the MSMs --
and never serve as i_k.  Deterministic in `seed`."""
import random
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def big_instance(path, n_cons, n_wires, n_pub, seed, bool_frac=0.1):
    """writes `path` (.r1cs) and returns the witness bytes (n_wires x 32, little-endian standard form)"""
    rng = random.Random(seed)
    gb = rng.getrandbits
    w = [1] + [0] * (n_wires - 1)
    nz = []                                         # wires that may be solved for (non-zero, not boolean)
    for i in range(1, n_wires):
        if rng.random() < bool_frac and i > n_pub:
            w[i] = gb(1)
        else:
            v = gb(253) or 1
            w[i] = v; nz.append(i)
    # batch inversion of the non-boolean wires
    pre = [1] * len(nz); acc = 1
    for j, i in enumerate(nz):
        pre[j] = acc; acc = acc * w[i] % R
    ia = pow(acc, R - 2, R); winv = {}
    for j in range(len(nz) - 1, -1, -1):
        i = nz[j]; winv[i] = ia * pre[j] % R; ia = ia * w[i] % R
    le = [x.to_bytes(32, 'little') for x in w]
    pk = struct.pack
    out = []; nnz = len(nz)
    for k in range(n_cons):
        if k % 7 == 0:
            c = 1 + gb(5); av = c; ab = pk('<II', 1, 0) + c.to_bytes(32, 'little')
        else:
            idx = sorted({gb(30) % n_wires for _ in range(1 + gb(2) % 3)}); av = 0; ab = [pk('<I', len(idx))]
            for i in idx:
                c = gb(253) or 1; av += c * w[i]; ab.append(pk('<I', i) + c.to_bytes(32, 'little'))
            av %= R; ab = b''.join(ab)
        idx = sorted({gb(30) % n_wires for _ in range(1 + gb(1))}); bv = 0; bb = [pk('<I', len(idx))]
        for i in idx:
            c = gb(253) or 1; bv += c * w[i]; bb.append(pk('<I', i) + c.to_bytes(32, 'little'))
        bv %= R
        i1 = nz[gb(30) % nnz]
        c1 = av * bv % R * winv[i1] % R
        out.append(ab); out.append(b''.join(bb))
        out.append(pk('<II', 1, i1) + c1.to_bytes(32, 'little') if c1 else pk('<I', 0))
    body = b''.join(out)
    hdr = pk('<I', 32) + R.to_bytes(32, 'little') + pk('<IIIIQI', n_wires, 0, n_pub, n_wires - 1 - n_pub, n_wires, n_cons)
    w2l = b''.join(pk('<Q', i) for i in range(n_wires))
    with open(path, 'wb') as f:
        f.write(b'r1cs' + pk('<II', 1, 3))
        for sid, data in ((1, hdr), (2, body), (3, w2l)):
            f.write(pk('<IQ', sid, len(data))); f.write(data)
    return b''.join(le)


# ---- a circuit-shaped instance: every constraint DEFINES a wire, so any assignment of the inputs has a witness ----
def _chain_rows(n_cons, n_in, seed):
    """row k: (A terms, B terms) over wires that exist before wire n_in + 1 + k; the row's C side is that new wire.  Structure only -- no witness values enter, so the
    writer and the witness generator below draw the same rows from the same seed."""
    rng = random.Random(seed); gb = rng.getrandbits
    for k in range(n_cons):
        hi = n_in + 1 + k                            # wires [0, hi) exist
        if k % 7 == 0:
            a = [(0, 1 + gb(5))]
        else:
            a = [(i, gb(253) or 1) for i in sorted({gb(30) % hi for _ in range(1 + gb(2) % 3)})]
        # half of the operands are recent wires (a chain: the witness cannot be computed out of order), half are anywhere
        b = [(i, gb(253) or 1) for i in sorted({(hi - 1 - gb(30) % min(hi, 64)) if gb(1) else gb(30) % hi for _ in range(1 + gb(1))})]
        yield a, b


def chain_instance(path, n_cons, n_in, n_pub, seed):
    """writes `path` (.r1cs): wires [1 | n_in inputs, the first n_pub of them public | one wire per constraint], constraint k: <A_k, w> . <B_k, w> = w[n_in + 1 + k]"""
    pk = struct.pack
    out = []
    for k, (a, b) in enumerate(_chain_rows(n_cons, n_in, seed)):
        out.append(pk('<I', len(a)) + b''.join(pk('<I', i) + c.to_bytes(32, 'little') for i, c in a))
        out.append(pk('<I', len(b)) + b''.join(pk('<I', i) + c.to_bytes(32, 'little') for i, c in b))
        out.append(pk('<II', 1, n_in + 1 + k) + (1).to_bytes(32, 'little'))
    n_wires = 1 + n_in + n_cons
    hdr = pk('<I', 32) + R.to_bytes(32, 'little') + pk('<IIIIQI', n_wires, 0, n_pub, n_wires - 1 - n_pub, n_wires, n_cons)
    w2l = b''.join(pk('<Q', i) for i in range(n_wires))
    with open(path, 'wb') as f:
        f.write(b'r1cs' + pk('<II', 1, 3))
        for sid, data in ((1, hdr), (2, b''.join(out)), (3, w2l)):
            f.write(pk('<IQ', sid, len(data))); f.write(data)
    return n_wires


def chain_witness(n_cons, n_in, seed, inputs):
    """the witness of chain_instance(.., seed) for the given n_in input values: forward evaluation, one product per constraint"""
    assert len(inputs) == n_in
    w = [1] + [x % R for x in inputs] + [0] * n_cons
    for k, (a, b) in enumerate(_chain_rows(n_cons, n_in, seed)):
        av = 0
        for i, c in a:
            av += c * w[i]
        bv = 0
        for i, c in b:
            bv += c * w[i]
        w[n_in + 1 + k] = (av % R) * (bv % R) % R
    return b''.join(x.to_bytes(32, 'little') for x in w)
