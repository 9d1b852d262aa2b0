"""R1CS of ZkFranchiseProofCircuit(nLevels) over the circom wire numbering (test-only setup support).

The reference produces `circuit.r1cs` with `circom census.circom --r1cs` (circuit/circuit-compiler.sh:91) and feeds it to
`snarkjs groth16 setup` (:112).  That blob is missing (.MISSING_LARGE_BLOBS:1) and circom is not available, so this module
restates the constraints of circuit/census.circom:49-115 and its circomlib 2.0.5 templates *over the surviving wires*
(every eliminated signal is substituted by its linear combination of wires, as circom -O2 does).  The constraint set
is this build's own -- equivalent in meaning, not claimed identical to circom's -- and is validated by
(a) A.w * B.w == C.w on reference-wasm witnesses and (b) rejection of mutated witnesses (tests/test_r1cs.py).

A linear combination is a dict {wire: coef mod r}; wire 0 is the constant one.
"""
import json
import os
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'poseidon_constants.json')
NROUNDSP = {3: 57, 4: 56, 5: 60}
_PC = None


def poseidon_params():
    global _PC
    if _PC is None:
        raw = json.load(open(_GOLD))
        _PC = {int(t): {k: [int(x, 16) for x in v] for k, v in a.items()} for t, a in raw.items()}
    return _PC


# ---------------- linear combinations ----------------
def lc_const(c):
    c %= R
    return {0: c} if c else {}


def lc_wire(w, c=1):
    return {w: c % R}


def lc_add(a, b, kb=1):
    """a + kb*b"""
    out = dict(a)
    kb %= R
    if kb == 0:
        return out
    for w, c in b.items():
        v = (out.get(w, 0) + kb * c) % R
        if v:
            out[w] = v
        else:
            out.pop(w, None)
    return out


def lc_scale(a, k):
    k %= R
    return {w: c * k % R for w, c in a.items()} if k else {}


def lc_is_const(a):
    return all(w == 0 for w in a)


def lc_cval(a):
    return a.get(0, 0)


def lc_eval(a, wit):
    return sum(c * wit[w] for w, c in a.items()) % R


def inv(x):
    return pow(x, -1, R)


class Layout:
    """Same numbers as csrc/zkc_device.h WitnessLayout::make (kept in step by tests/test_r1cs.py)."""
    kHash3 = 20 + 2 + 57 + 46 + 114
    kHash1New = 26 + 3 + 56 + 60 + 112
    kSik = 27 + 3 + 56 + 62 + 112
    kNullifier = 296

    def __init__(self, nL):
        self.nL, self.n = nL, nL + 1
        n = self.n
        o = self.lvl_off(n - 1) + 2
        self.off_n2bnew = o; o += (253 - n) + 127 + 133
        self.off_n2bold = o; o += 253 + 127 + 133
        self.off_levins = o; o += n - 2
        self.off_iszero = o; o += 2 * (n - 2) + 1
        o += 1
        self.ver_size = o
        self.off_census = 13 + 2 * nL
        self.off_checknull = self.off_census + self.ver_size
        self.off_checkweight = self.off_checknull + 1
        self.off_nullifier = self.off_checkweight + 251
        self.off_sik = self.off_nullifier + self.kNullifier
        self.off_sikver = self.off_sik + 1 + self.kSik
        self.nWires = self.off_sikver + self.ver_size
        self.nInputs = 12 + 2 * n

    def lvl_off(self, i):
        base = 4 + self.kHash1New
        if i == 0:
            return base
        o = base + (5 + self.kHash3) + (i - 1) * (6 + self.kHash3)
        if i > self.n - 3:
            o += 1
        if i > self.n - 2:
            o -= 1
        return o


class R1CS:
    def __init__(self, nWires, nPub):
        self.nWires, self.nPub = nWires, nPub
        self.cons = []          # (A, B, C) dicts

    def add(self, a, b, c):
        self.cons.append((a, b, c))

    def check(self, wit):
        """first violated constraint index or -1"""
        for k, (a, b, c) in enumerate(self.cons):
            if (lc_eval(a, wit) * lc_eval(b, wit) - lc_eval(c, wit)) % R:
                return k
        return -1

    def write(self, path):
        """iden3 .r1cs binary (r1csfile 0.0.45 layout: header, constraints, wire2label)."""
        def lc_bytes(a):
            items = sorted(a.items())
            return struct.pack('<I', len(items)) + b''.join(struct.pack('<I', w) + c.to_bytes(32, 'little') for w, c in items)
        body = b''.join(lc_bytes(a) + lc_bytes(b) + lc_bytes(c) for a, b, c in self.cons)
        hdr = struct.pack('<I', 32) + R.to_bytes(32, 'little') + struct.pack('<IIIIQI', self.nWires, 0, self.nPub,
                                                                           self.nWires - 1 - self.nPub, self.nWires, len(self.cons))
        w2l = b''.join(struct.pack('<Q', i) for i in range(self.nWires))
        with open(path, 'wb') as f:
            f.write(b'r1cs' + struct.pack('<II', 1, 3))
            for sid, data in ((1, hdr), (2, body), (3, w2l)):
                f.write(struct.pack('<IQ', sid, len(data))); f.write(data)


# ---------------- Poseidon templates ----------------
def _sigma(cs, x, w_in2, out_lc):
    """x^5 = out with wires in2 = w_in2, in4 = w_in2 + 1"""
    i2, i4 = lc_wire(w_in2), lc_wire(w_in2 + 1)
    cs.add(x, x, i2); cs.add(i2, i2, i4); cs.add(i4, x, out_lc)


def _mix(st, M, t):
    out = []
    for i in range(t):
        acc = {}
        for j in range(t):
            acc = lc_add(acc, st[j], M[j * t + i])
        out.append(acc)
    return out


def poseidon_std(cs, ins, blk, out_wire, cmask):
    """t = len(ins)+1 in {3,4}; ins = input LCs; survivor layout 0 (csrc/zkc_witness.hip poseidon_trace<T,0>)."""
    t = len(ins) + 1
    pp = poseidon_params()[t]; C, S, M, P = pp['C'], pp['S'], pp['M'], pp['P']; RP = NROUNDSP[t]
    rank, nc1 = [], 0
    for j in range(t):
        rank.append(nc1); nc1 += 0 if (cmask >> j) & 1 else 1
    nA = nc1 + 6 * t; oLast = nA; oMS = nA + t - 1; oF = oMS + RP; oP = oF + 2 * (nc1 + 7 * t)
    ark_idx = lambda r, j: (rank[j] if r == 1 else nc1 + (r - 2) * t + j)
    sF_idx = lambda r, j: oF + 2 * (rank[j] if r == 0 else nc1 + (r - 1) * t + j)
    st = [lc_const(C[0])] + [lc_add(lc_const(C[j]), ins[j - 1]) for j in range(1, t)]
    for r in range(4):
        ns = []
        for j in range(t):
            cj = C[(r + 1) * t + j]
            if lc_is_const(st[j]):
                assert r == 0 and (cmask >> j) & 1
                ns.append(lc_const(pow(lc_cval(st[j]), 5, R) + cj))
            else:
                assert not (r == 0 and (cmask >> j) & 1)
                aw = lc_wire(blk + ark_idx(r + 1, j))
                _sigma(cs, st[j], blk + sF_idx(r, j), lc_add(aw, lc_const(cj), -1))
                ns.append(aw)
        st = _mix(ns, M if r < 3 else P, t)
    for r in range(RP):
        Sr = S[(2 * t - 1) * r:(2 * t - 1) * (r + 1)]
        m = lc_wire(blk + oMS + r)
        rest = {}
        for i in range(1, t):
            rest = lc_add(rest, st[i], Sr[i])
        in0 = lc_scale(lc_add(m, rest, -1), inv(Sr[0]))                  # mixS[r].in[0]
        _sigma(cs, st[0], blk + oP + 2 * r, lc_add(in0, lc_const(C[5 * t + r]), -1))
        st = [m] + [lc_add(st[i], in0, Sr[t + i - 1]) for i in range(1, t)]
    for r in range(3):
        ns = []
        for j in range(t):
            aw = lc_wire(blk + ark_idx(5 + r, j))
            _sigma(cs, st[j], blk + sF_idx(4 + r, j), lc_add(aw, lc_const(C[5 * t + RP + r * t + j]), -1))
            ns.append(aw)
        st = _mix(ns, M, t)
    last = {}
    for j in range(t - 1):
        _sigma(cs, st[j], blk + sF_idx(7, j), lc_wire(blk + oLast + j))
        last = lc_add(last, lc_wire(blk + oLast + j), M[j * t])
    out_last = lc_scale(lc_add(lc_wire(out_wire), last, -1), inv(M[(t - 1) * t]))
    _sigma(cs, st[t - 1], blk + sF_idx(7, t - 1), out_last)
    return oP + 2 * RP


def poseidon_t5(cs, ins, blk, out_wire):
    """The single t=5 instance (computedNullifier): survivor layout 1 (poseidon_trace<5,1>)."""
    t = 5
    pp = poseidon_params()[t]; C, S, M, P = pp['C'], pp['S'], pp['M'], pp['P']; RP = 60
    W = lambda k: lc_wire(blk + k)
    ark_idx = lambda r, j: (j - 1 if r == 1 else 4 + (r - 2) * 5 + j if r <= 3 else 14 if r == 4 else 15 + (r - 5) * 5 + j)
    sF_idx = lambda r, j: 98 + 2 * ((j - 1) if r == 0 else 4 + (r - 1) * 5 + j)
    Sp = lambda r, i: S[9 * r + t + i - 1]          # mixS[r]: out[i] = in[i] + in[0]*Sp(r,i)
    st = [lc_const(C[0])] + [lc_add(lc_const(C[j]), ins[j - 1]) for j in range(1, t)]
    for r in range(3):
        ns = []
        for j in range(t):
            cj = C[(r + 1) * t + j]
            if lc_is_const(st[j]):
                ns.append(lc_const(pow(lc_cval(st[j]), 5, R) + cj))
            else:
                aw = W(ark_idx(r + 1, j))
                _sigma(cs, st[j], blk + sF_idx(r, j), lc_add(aw, lc_const(cj), -1))
                ns.append(aw)
        st = _mix(ns, M, t)
    x3 = st                                           # inputs of sigmaF[3][*]
    # in0_r for every partial round from the tracked out[4] wires
    q = lambda r: W(30) if r < 0 else W(35 + r)       # mixS[r].out[4], r <= 56 ; q(-1) = mix[3].out[4]
    in0 = [None] * RP
    for r in range(57):
        in0[r] = lc_scale(lc_add(q(r), q(r - 1), -1), inv(Sp(r, 4)))
    in0[57] = lc_scale(lc_add(W(95), q(56), -1), inv(Sp(57, 4)))
    in0[58] = lc_scale(lc_add(W(96), W(95), -1), inv(Sp(58, 4)))
    in0[59] = W(97)
    # s[r][i]: state word i entering partial round r
    s = [[None] * t for _ in range(RP + 1)]
    for i in (1, 2, 3):
        s[58][i] = W(92 + i - 1)
        for r in range(57, -1, -1):
            s[r][i] = lc_add(s[r + 1][i], in0[r], -Sp(r, i))
        s[59][i] = lc_add(s[58][i], in0[58], Sp(58, i))
        s[60][i] = lc_add(s[59][i], in0[59], Sp(59, i))
    for r in range(58):
        s[r][4] = q(r - 1)
    s[58][4] = W(95); s[59][4] = W(96); s[60][4] = lc_add(W(96), in0[59], Sp(59, 4))
    # solve a_1..a_4 (ark[4].out) from s[0][1..4] = sum_i P[i][j] a_i with a_0 a wire
    a0 = W(14)
    N = [[P[i * t + j] for i in range(1, 5)] for j in range(1, 5)]
    rhs = [lc_add(s[0][j], a0, -P[0 * t + j]) for j in range(1, 5)]
    # Gauss-Jordan over Fr on the 4x4 system, carrying LC right-hand sides
    for col in range(4):
        piv = next(r_ for r_ in range(col, 4) if N[r_][col])
        N[col], N[piv] = N[piv], N[col]; rhs[col], rhs[piv] = rhs[piv], rhs[col]
        iv = inv(N[col][col])
        N[col] = [x * iv % R for x in N[col]]; rhs[col] = lc_scale(rhs[col], iv)
        for r_ in range(4):
            if r_ != col and N[r_][col]:
                f = N[r_][col]
                N[r_] = [(x - f * y) % R for x, y in zip(N[r_], N[col])]
                rhs[r_] = lc_add(rhs[r_], rhs[col], -f)
    a = [a0] + rhs
    for j in range(t):
        _sigma(cs, x3[j], blk + sF_idx(3, j), lc_add(a[j], lc_const(C[4 * t + j]), -1))
    s0 = {}
    for i in range(t):
        s0 = lc_add(s0, a[i], P[i * t + 0])
    s[0][0] = s0
    for r in range(RP):
        Sr = S[9 * r:9 * r + 9]
        _sigma(cs, s[r][0], blk + 176 + 2 * r, lc_add(in0[r], lc_const(C[5 * t + r]), -1))
        n0 = lc_scale(in0[r], Sr[0])
        for i in range(1, t):
            n0 = lc_add(n0, s[r][i], Sr[i])
        s[r + 1][0] = n0
    st = s[RP]
    for r in range(3):
        ns = []
        for j in range(t):
            aw = W(ark_idx(5 + r, j))
            _sigma(cs, st[j], blk + sF_idx(4 + r, j), lc_add(aw, lc_const(C[5 * t + RP + r * t + j]), -1))
            ns.append(aw)
        st = _mix(ns, M, t)
    last = {}
    for j in range(t - 1):
        _sigma(cs, st[j], blk + sF_idx(7, j), W(31 + j))
        last = lc_add(last, W(31 + j), M[j * t])
    _sigma(cs, st[t - 1], blk + sF_idx(7, t - 1), lc_scale(lc_add(lc_wire(out_wire), last, -1), inv(M[(t - 1) * t])))


# ---------------- bit decompositions ----------------
def _bool(cs, b):
    cs.add(b, lc_add(b, lc_const(1), -1), {})


def num2bits(cs, value_lc, bit_lcs, nbits):
    """bit_lcs: {i: LC} for the bits that are wires (or constants); exactly one index `top` missing -> solved linearly."""
    missing = [i for i in range(nbits) if i not in bit_lcs]
    assert len(missing) == 1
    top = missing[0]
    acc = dict(value_lc)
    for i, b in bit_lcs.items():
        acc = lc_add(acc, b, -(1 << i))
    bit_lcs[top] = lc_scale(acc, inv(1 << top))
    for i in range(nbits):
        if not lc_is_const(bit_lcs[i]):
            _bool(cs, bit_lcs[i])
        else:
            assert lc_cval(bit_lcs[i]) in (0, 1)
    return bit_lcs


def alias_check(cs, bits, w_parts):
    """AliasCheck -> CompConstant(r-1): parts wires w_parts..+126, Num2Bits(135) bits wires after them (127 and 134 missing)."""
    ct = R - 1
    b = (1 << 128) - 1; a = 1; e = 1
    sout = {}
    for i in range(127):
        clsb, cmsb = (ct >> (2 * i)) & 1, (ct >> (2 * i + 1)) & 1
        sl, sm = bits[2 * i], bits[2 * i + 1]
        p = lc_wire(w_parts + i)
        if cmsb == 0 and clsb == 0:      # parts = -b sm sl + b sm + b sl
            cs.add(lc_scale(sm, -b), sl, lc_add(lc_add(p, sm, -b), sl, -b))
        elif cmsb == 0 and clsb == 1:    # parts = a sm sl - a sl + b sm - a sm + a
            cs.add(lc_scale(sm, a), sl, lc_add(lc_add(lc_add(p, sl, a), sm, a - b), lc_const(a), -1))
        elif cmsb == 1 and clsb == 0:    # parts = b sm sl - a sm + a
            cs.add(lc_scale(sm, b), sl, lc_add(lc_add(p, sm, a), lc_const(a), -1))
        else:                            # parts = -a sm sl + a
            cs.add(lc_scale(sm, -a), sl, lc_add(p, lc_const(a), -1))
        sout = lc_add(sout, p)
        b -= e; a += e; e *= 2
    w = w_parts + 127
    bl = {}
    for i in range(134):
        if i == 127:
            bl[i] = {}                   # compConstant.out === 0
        else:
            bl[i] = lc_wire(w); w += 1
    num2bits(cs, sout, bl, 135)


def is_zero(cs, x, out_lc, inv_wire):
    cs.add(x, lc_wire(inv_wire), lc_add(lc_const(1), out_lc, -1))
    cs.add(x, out_lc, {})


def tautology(cs, w):
    cs.add(lc_wire(w), lc_const(1), lc_wire(w))


# ---------------- SMTVerifier ----------------
def smt_verifier(cs, L, blk, key_w, value_lc, root_w, sib_w0):
    n = L.n
    key = lc_wire(key_w)
    sib = [lc_wire(sib_w0 + i) for i in range(n - 1)] + [{}]          # siblings[n-1] is forced to 0 and is not a wire
    one = lc_const(1)
    is_zero(cs, key, lc_wire(blk + 0), blk + 1)                        # areKeyEquals (oldKey = 0)
    tautology(cs, blk + 2)                                             # checkRoot.isz.inv: unconstrained by the circuit
    h1 = lc_wire(blk + 3)
    poseidon_std(cs, [key, value_lc, one], blk + 4, blk + 3, 1 | 8)
    # wires of the level blocks
    def lvl(i):
        lb = blk + L.lvl_off(i); o = 0; d = {}
        if i == n - 1:
            d['T'] = lb
            if n <= 253:
                d['bit'] = lb + 1        # n = 254 (nLevels = 253, the largest circomlib allows): all 254 key bits steer a level, so the one bit Num2Bits' linear
            return d                     # constraint solves for is this level's; it is not a wire and the alias check's wires start one place earlier (off_n2bnew - 1)
        if i == n - 3:
            d['T'] = lb + o; o += 1
        if 0 < i < n - 2:
            d['N'] = lb + o; o += 1
        d['bit'] = lb + o; d['child'] = lb + o + 1; d['aux0'] = lb + o + 2; d['out'] = lb + o + 3; d['L'] = lb + o + 4; d['hash'] = lb + o + 5
        return d
    lv = [lvl(i) for i in range(n)]
    # key bits: lrbit wires for 0..n-1, n2bNew.out wires for n..252, bit 253 solved
    bits = {i: lc_wire(lv[i]['bit']) for i in range(n) if 'bit' in lv[i]}
    for i in range(n, 253):
        bits[i] = lc_wire(blk + L.off_n2bnew + i - n)
    num2bits(cs, key, bits, 254)
    alias_check(cs, bits, blk + L.off_n2bnew + (253 - n))
    ob = {i: lc_wire(blk + L.off_n2bold + i) for i in range(253)}
    num2bits(cs, {}, ob, 254)
    alias_check(cs, ob, blk + L.off_n2bold + 253)
    # SMTLevIns / SMTVerifierSM
    V = {i: lc_wire(blk + L.off_levins + i - 1) for i in range(1, n - 1)}
    Nw = {i: lc_wire(lv[i]['N']) for i in range(1, n - 2)}
    T_n3, T_n2 = lc_wire(lv[n - 3]['T']), lc_wire(lv[n - 1]['T'])
    N = dict(Nw)
    N0 = lc_add(one, T_n3, -1)
    for i in range(1, n - 2):
        N0 = lc_add(N0, Nw[i], -1)
    N[0] = N0; N[n - 2] = lc_add(T_n3, T_n2, -1); N[n - 1] = T_n2
    V[0] = N0
    Vn1 = lc_add(one, N0, -1)
    for i in range(1, n - 1):
        Vn1 = lc_add(Vn1, V[i], -1)
    V[n - 1] = Vn1
    T = {-1: one}
    for i in range(n):
        T[i] = lc_add(T[i - 1], N[i], -1)
    T[n - 3] = T_n3; T[n - 2] = T_n2; T[n - 1] = {}
    oz = blk + L.off_iszero
    Z = {i: lc_wire(oz + 2 * i) for i in range(n - 2)}
    Z[n - 2] = lc_add(one, V[n - 1], -1)
    for i in range(n - 1):
        is_zero(cs, sib[i], Z[i], oz + 2 * i + 1 if i < n - 2 else oz + 2 * (n - 2))
    tautology(cs, oz + 2 * (n - 2) + 1)                                 # isZero[n-1].inv
    D = {n - 2: V[n - 1]}
    for i in range(n - 2, 0, -1):
        cs.add(lc_add(one, D[i], -1), lc_add(one, Z[i - 1], -1), V[i])  # levIns[i] = (1-done[i])(1-isZero[i-1].out)
        D[i - 1] = lc_add(D[i], V[i])
    for i in range(1, n):
        cs.add(T[i - 1], V[i], N[i])                                    # st_inew[i] = prev_top * levIns[i]
    # levels
    for i in range(n - 2, -1, -1):
        d = lv[i]
        child, L_, b = lc_wire(d['child']), lc_wire(d['L']), bits[i]
        if i == n - 2:
            cs.add(h1, N[n - 1], child)                                 # root[n-1] = H(0,0)*st_top[n-1](=0) + hash1New*st_inew[n-1]
        cs.add(lc_add(sib[i], child, -1), b, lc_add(L_, child, -1))     # Switcher
        Rr = lc_add(lc_add(sib[i], child), L_, -1)
        poseidon_std(cs, [L_, Rr], d['hash'], d['out'], 1)
        cs.add(lc_wire(d['out']), T[i], lc_wire(d['aux0']))             # aux[0] = proofHash.out * st_top
        parent = lc_wire(lv[i - 1]['child']) if i > 0 else lc_wire(root_w)
        cs.add(h1, N[i], lc_add(parent, lc_wire(d['aux0']), -1))        # root = aux[0] + new1leaf * st_inew


def build(nLevels=160):
    L = Layout(nLevels)
    cs = R1CS(L.nWires, 8)
    one = lc_const(1)
    W = lc_wire
    # wires: 1,2 electionId; 3 nullifier; 4,5 voteHash; 6 sikRoot; 7 censusRoot; 8 voteWeight; 9 availableWeight;
    #        10 address; 11 password; 12 signature; 13.. censusSiblings; 13+nL.. sikSiblings
    smt_verifier(cs, L, L.off_census, 10, W(9), 7, 13)
    tautology(cs, L.off_checknull)                                      # checkNullifier.isz.inv
    # checkWeight = LessEqThan(252)(voteWeight, availableWeight): Num2Bits(253)(voteWeight + 2^252 - availableWeight - 1), bit 252 == 0
    x = lc_add(lc_add(W(8), W(9), -1), lc_const((1 << 252) - 1))
    bl = {i: W(L.off_checkweight + i) for i in range(251)}
    bl[252] = {}
    num2bits(cs, x, bl, 253)
    poseidon_t5(cs, [W(12), W(11), W(1), W(2)], L.off_nullifier, 3)
    poseidon_std(cs, [W(10), W(11), W(12)], L.off_sik + 1, L.off_sik, 1)
    smt_verifier(cs, L, L.off_sikver, 10, W(L.off_sik), 6, 13 + nLevels)
    # voteHash (wires 4,5) is deliberately unconstrained (census.circom:54-57)
    return L, cs
