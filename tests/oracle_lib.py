"""ctypes binding of the CPU oracle (oracle/_build/libzkc_oracle.so).  TEST INFRASTRUCTURE: imported only from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the product package."""
import ctypes, json, os, subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
SO = os.path.join(ORACLE_DIR, '_build', 'libzkc_oracle.so')
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
INPUT_KEYS = ['electionId', 'nullifier', 'availableWeight', 'voteHash', 'sikRoot', 'censusRoot', 'address', 'password',
              'signature', 'voteWeight', 'censusSiblings', 'sikSiblings']
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = ctypes.CDLL(SO)
        _lib.zko_init()
        _lib.zko_msm_g1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib.zko_msm_g2.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
        _lib.zko_groth16_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p,
                                           ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p]
        _lib.zko_zkey_vk.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        _lib.zko_zkey_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    return _lib


def le32(x):
    return int(x).to_bytes(32, 'little')


def flat_inputs(inp, nLevels=160):
    """12-key circuit input object (decimal strings; sibling lists may be shorter than nLevels+1) -> 32-byte LE block
    in census.circom declaration order (reduced mod r like snarkjs/circom_runtime do)."""
    out = []
    for k in INPUT_KEYS:
        v = inp[k]
        if k.endswith('Siblings'):
            v = list(v) + ['0'] * (nLevels + 1 - len(v))
            assert len(v) == nLevels + 1
        out += [int(x) % R for x in v] if isinstance(v, list) else [int(v) % R]
    return b''.join(le32(x) for x in out)


def poseidon(xs):
    buf = b''.join(le32(x) for x in xs)
    out = ctypes.create_string_buffer(32)
    lib().zko_poseidon(out, buf, len(xs))
    return int.from_bytes(out.raw, 'little')


def witness(inp_or_flat, nLevels=160):
    L = lib()
    buf = inp_or_flat if isinstance(inp_or_flat, (bytes, bytearray)) else flat_inputs(inp_or_flat, nLevels)
    nw = L.zko_n_wires(nLevels)
    out = ctypes.create_string_buffer(nw * 32)
    rc = L.zko_witness(nLevels, buf, out)
    return rc, out.raw


def g1_json(p):
    return le32(p[0]) + le32(p[1]) if int(p[2]) != 0 else bytes(64)


def g2_json(p):
    if int(p[2][0]) == 0 and int(p[2][1]) == 0:
        return bytes(128)
    return le32(p[0][0]) + le32(p[0][1]) + le32(p[1][0]) + le32(p[1][1])


def vk_bytes(vk):
    b = g1_json(vk['vk_alpha_1']) + g2_json(vk['vk_beta_2']) + g2_json(vk['vk_gamma_2']) + g2_json(vk['vk_delta_2'])
    for p in vk['IC']:
        b += g1_json(p)
    return b


def proof_bytes(pr):
    return g1_json(pr['pi_a']) + g2_json(pr['pi_b']) + g1_json(pr['pi_c'])


def proof_json(pb):
    i = lambda o: str(int.from_bytes(pb[o:o + 32], 'little'))
    return {'pi_a': [i(0), i(32), '1'], 'pi_b': [[i(64), i(96)], [i(128), i(160)], ['1', '0']], 'pi_c': [i(192), i(224), '1'],
            'protocol': 'groth16', 'curve': 'bn128'}


def verify(vk, pub, proof):
    """vk/proof: parsed JSON dicts or raw bytes; pub: list of decimal strings/ints or raw bytes."""
    vkb = vk if isinstance(vk, (bytes, bytearray)) else vk_bytes(vk)
    pb = proof if isinstance(proof, (bytes, bytearray)) else proof_bytes(proof)
    pubb = pub if isinstance(pub, (bytes, bytearray)) else b''.join(le32(x) for x in pub)
    npub = len(pubb) // 32
    assert len(vkb) == 448 + 64 * (npub + 1)
    return bool(lib().zko_groth16_verify(vkb, npub, pubb, pb))


def ntt(vals, inverse=False):
    n = len(vals); logn = n.bit_length() - 1
    buf = ctypes.create_string_buffer(b''.join(le32(v) for v in vals), n * 32)
    lib().zko_ntt(buf, logn, 1 if inverse else 0)
    raw = buf.raw                                   # one copy (buf.raw inside the loop copied the whole buffer per element)
    return [int.from_bytes(raw[32 * i:32 * i + 32], 'little') for i in range(n)]


def root_of_unity(logn):
    out = ctypes.create_string_buffer(32)
    lib().zko_root_of_unity(out, logn)
    return int.from_bytes(out.raw, 'little')


def msm_g1(bases, scalars):
    out = ctypes.create_string_buffer(64)
    lib().zko_msm_g1(out, bytes(bases), bytes(scalars), len(scalars) // 32)
    return out.raw


def msm_g2(bases, scalars):
    out = ctypes.create_string_buffer(128)
    lib().zko_msm_g2(out, bytes(bases), bytes(scalars), len(scalars) // 32)
    return out.raw


def g1_mul(base, k):
    out = ctypes.create_string_buffer(64)
    lib().zko_g1_mul(out, base, le32(k))
    return out.raw


def prove(zkey, wtns, r, s, npub=8):
    proof = ctypes.create_string_buffer(256); pub = ctypes.create_string_buffer(32 * npub)
    rc = lib().zko_groth16_prove(zkey, len(zkey), wtns, len(wtns) // 32, le32(r), le32(s), proof, pub)
    return rc, proof.raw, pub.raw


def zkey_vk(zkey, npub=8):
    out = ctypes.create_string_buffer(448 + 64 * (npub + 1))
    rc = lib().zko_zkey_vk(zkey, len(zkey), out)
    assert rc == 0, rc
    return out.raw


class ZKey(ctypes.Structure):
    _fields_ = [('nVars', ctypes.c_uint32), ('nPublic', ctypes.c_uint32), ('domainSize', ctypes.c_uint32), ('nCoeffs', ctypes.c_uint32)] + \
               [(n, ctypes.c_void_p) for n in ('alpha1', 'beta1', 'beta2', 'gamma2', 'delta1', 'delta2', 'ic', 'coeffs', 'pointsA',
                                                'pointsB1', 'pointsB2', 'pointsC', 'pointsH')]


def zkey_parse(zkey):
    z = ZKey()
    rc = lib().zko_zkey_parse(zkey, len(zkey), ctypes.byref(z))
    assert rc == 0, rc
    return z


def build_abc(zkey, wtns):
    z = zkey_parse(zkey); n = z.domainSize
    A = ctypes.create_string_buffer(32 * n); B = ctypes.create_string_buffer(32 * n); C = ctypes.create_string_buffer(32 * n)
    lib().zko_build_abc(ctypes.byref(z), wtns, A, B, C)
    return A.raw, B.raw, C.raw


def h_evals(zkey, wtns):
    z = zkey_parse(zkey); n = z.domainSize
    P = ctypes.create_string_buffer(32 * n)
    lib().zko_h_evals(ctypes.byref(z), wtns, P)
    return P.raw


def pmap(fn, items, workers=None):
    """fn over items on host threads (every oracle entry point is a ctypes call that releases the GIL): the GPU suite's oracle legs are tens to hundreds of CPU proofs and ran
    one after the other until round 4 -- most of the suite's wall time.  Order of results = order of items."""
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    if not items:
        return []
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0)), len(items)))
    lib()
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


def golden(name):
    return os.path.join(ROOT, 'tests', 'golden', name)


def load_json(name):
    return json.load(open(golden(name)))


def status_of_wasm_message(msg):
    """The per-voter status (include/zkcensus.h ZKC_W_*) that names the assert in a message of the reference's witness calculator: the outermost frame gives
    the census.circom line, the innermost tells SMTLevIns (last sibling) from ForceEqualIfEnabled (root / nullifier)."""
    import re
    frames = re.findall(r'Error in template (\w+?)_\d+ line: (\d+)', msg)
    assert frames and frames[-1][0] == 'ZkFranchiseProofCircuit', msg
    line, inner = int(frames[-1][1]), frames[0][0]
    return {(72, 'ZkFranchiseProofCircuit'): 1, (90, 'SMTLevIns'): 7, (90, 'ForceEqualIfEnabled'): 2, (103, 'SMTLevIns'): 5,
            (103, 'ForceEqualIfEnabled'): 3, (114, 'ForceEqualIfEnabled'): 4}[(line, inner)]


def twist_point_outside_g2(start=1):
    """A point on the twist y^2 = x^3 + 3/(9 + u) over Fq2 that is (almost surely) NOT in the order-r subgroup G2: the first x = x0 + u, x0 >= start, whose right-hand side
    is a square.  Returns ((x0, x1), (y0, y1)) as integers.  Used to check that the verifiers refuse a proof whose B leaves G2 (the twist has a cofactor)."""
    q = Q
    def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
    def f2pow(a, e):
        r = (1, 0)
        while e:
            if e & 1: r = f2mul(r, a)
            a = f2mul(a, a); e >>= 1
        return r
    def f2inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, q); return (a[0] * n % q, -a[1] * n % q)
    def f2sqrt(a):                                         # q = 3 mod 4: complex method
        if a == (0, 0): return a
        a1 = f2pow(a, (q - 3) // 4); alpha = f2mul(f2mul(a1, a1), a); x0 = f2mul(a1, a)
        if alpha == (q - 1, 0): return f2mul((0, 1), x0)
        b = f2pow(((1 + alpha[0]) % q, alpha[1]), (q - 1) // 2); return f2mul(b, x0)
    Btw = f2mul((3, 0), f2inv((9, 1)))
    for x0 in range(start, start + 200):
        x = (x0, 1); rhs = f2mul(f2mul(x, x), x); rhs = ((rhs[0] + Btw[0]) % q, (rhs[1] + Btw[1]) % q)
        y = f2sqrt(rhs)
        if f2mul(y, y) == rhs:
            return (x, y)
    raise AssertionError('no twist point found')
