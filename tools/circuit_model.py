"""Exploration model (pure Python, big ints) of census.circom + the circomlib 2.0.5 templates it pulls in
(reference: circuit/census.circom:49-115; circomlib templates restated from their published definitions,
SURVEY.md Appendix C).  It evaluates EVERY template signal and records (hierarchical name, value) so that the
wire order circom 2.1.5 chose can be recovered by value-matching against the wasm oracle
(tools/derive_wire_map.py).  Not part of the product or the test-suite."""
import json, os

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
_C = json.load(open(os.path.join(os.path.dirname(__file__), '..', 'zk-franchise-proof-circuit_amd/data/poseidon_constants.json')))
PC = {int(t): {k: [int(x, 16) for x in v] for k, v in a.items()} for t, a in _C.items()}
NROUNDSP = {3: 57, 4: 56, 5: 60}

class Rec:
    def __init__(self): self.names = []; self.vals = []
    def s(self, name, v):
        v %= R; self.names.append(name); self.vals.append(v); return v

def sigma(rec, pfx, x):
    rec.s(pfx + '.in', x)
    in2 = rec.s(pfx + '.in2', x * x)
    in4 = rec.s(pfx + '.in4', in2 * in2)
    return rec.s(pfx + '.out', in4 * x)

def poseidon(rec, pfx, inputs):
    t = len(inputs) + 1
    C, S, M, P = PC[t]['C'], PC[t]['S'], PC[t]['M'], PC[t]['P']
    RP = NROUNDSP[t]
    for i, x in enumerate(inputs): rec.s('%s.inputs[%d]' % (pfx, i), x)
    p = pfx + '.pEx'
    def ark(r, st, off):
        for j in range(t): rec.s('%s.ark[%d].in[%d]' % (p, r, j), st[j])
        return [rec.s('%s.ark[%d].out[%d]' % (p, r, j), st[j] + C[off + j]) for j in range(t)]
    def mix(r, st, MM):
        for j in range(t): rec.s('%s.mix[%d].in[%d]' % (p, r, j), st[j])
        return [rec.s('%s.mix[%d].out[%d]' % (p, r, i), sum(MM[j * t + i] * st[j] for j in range(t))) for i in range(t)]
    st = ark(0, [0] + list(inputs), 0)
    for r in range(3):
        st = [sigma(rec, '%s.sigmaF[%d][%d]' % (p, r, j), st[j]) for j in range(t)]
        st = ark(r + 1, st, (r + 1) * t)
        st = mix(r, st, M)
    st = [sigma(rec, '%s.sigmaF[3][%d]' % (p, j), st[j]) for j in range(t)]
    st = ark(4, st, 4 * t)
    st = mix(3, st, P)
    for r in range(RP):
        o = sigma(rec, '%s.sigmaP[%d]' % (p, r), st[0])
        ins = [rec.s('%s.mixS[%d].in[0]' % (p, r), o + C[5 * t + r])] + [rec.s('%s.mixS[%d].in[%d]' % (p, r, j), st[j]) for j in range(1, t)]
        out0 = rec.s('%s.mixS[%d].out[0]' % (p, r), sum(S[(2 * t - 1) * r + i] * ins[i] for i in range(t)))
        st = [out0] + [rec.s('%s.mixS[%d].out[%d]' % (p, r, i), ins[i] + ins[0] * S[(2 * t - 1) * r + t + i - 1]) for i in range(1, t)]
    for r in range(3):
        st = [sigma(rec, '%s.sigmaF[%d][%d]' % (p, 4 + r, j), st[j]) for j in range(t)]
        st = ark(5 + r, st, 5 * t + RP + r * t)
        st = mix(4 + r, st, M)
    st = [sigma(rec, '%s.sigmaF[7][%d]' % (p, j), st[j]) for j in range(t)]
    for j in range(t): rec.s('%s.mixLast[0].in[%d]' % (p, j), st[j])
    out = rec.s('%s.mixLast[0].out' % p, sum(M[j * t + 0] * st[j] for j in range(t)))
    rec.s(p + '.out[0]', out)
    return rec.s(pfx + '.out', out)

def is_zero(rec, pfx, x):
    x %= R
    rec.s(pfx + '.in', x)
    inv = rec.s(pfx + '.inv', pow(x, -1, R) if x else 0)
    return rec.s(pfx + '.out', 1 - x * inv)

def num2bits(rec, pfx, x, n):
    x %= R
    rec.s(pfx + '.in', x)
    return [rec.s('%s.out[%d]' % (pfx, i), (x >> i) & 1) for i in range(n)]

def comp_constant(rec, pfx, bits, ct):
    b = (1 << 128) - 1; a = 1; e = 1; total = 0
    for i in range(127):
        clsb = (ct >> (2 * i)) & 1; cmsb = (ct >> (2 * i + 1)) & 1
        sl, sm = bits[2 * i], bits[2 * i + 1]
        if cmsb == 0 and clsb == 0: v = -b * sm * sl + b * sm + b * sl
        elif cmsb == 0 and clsb == 1: v = a * sm * sl - a * sl + b * sm - a * sm + a
        elif cmsb == 1 and clsb == 0: v = b * sm * sl - a * sm + a
        else: v = -a * sm * sl + a
        total += rec.s('%s.parts[%d]' % (pfx, i), v)
        b -= e; a += e; e *= 2
    sout = rec.s(pfx + '.sout', total)
    nb = num2bits(rec, pfx + '.num2bits', sout, 135)
    return rec.s(pfx + '.out', nb[127])

def num2bits_strict(rec, pfx, x):
    rec.s(pfx + '.in', x)
    bits = num2bits(rec, pfx + '.n2b', x, 254)
    for i in range(254): rec.s('%s.out[%d]' % (pfx, i), bits[i])
    comp_constant(rec, pfx + '.aliasCheck.compConstant', bits, R - 1)
    return bits

def smt_verifier(rec, pfx, n, root, siblings, key, value):
    enabled = 1; fnc = 0
    rec.s(pfx + '.root', root); rec.s(pfx + '.key', key); rec.s(pfx + '.value', value)
    for i in range(n): rec.s('%s.siblings[%d]' % (pfx, i), siblings[i])
    h1old = poseidon(rec, pfx + '.hash1Old.h', [0, 0, 1])
    h1new = poseidon(rec, pfx + '.hash1New.h', [key, value, 1])
    rec.s(pfx + '.hash1New.out', h1new); rec.s(pfx + '.hash1New.key', key); rec.s(pfx + '.hash1New.value', value)
    bits = num2bits_strict(rec, pfx + '.n2bNew', key)
    num2bits_strict(rec, pfx + '.n2bOld', 0)   # oldKey = 0: bits are still wires (circom cannot fold them)
    # SMTLevIns
    p = pfx + '.smtLevIns'
    for i in range(n): rec.s('%s.siblings[%d]' % (p, i), siblings[i])
    isz = [is_zero(rec, '%s.isZero[%d]' % (p, i), siblings[i]) for i in range(n)]
    lev = [0] * n; done = [0] * (n - 1)
    lev[n - 1] = rec.s('%s.levIns[%d]' % (p, n - 1), 1 - isz[n - 2])
    done[n - 2] = rec.s('%s.done[%d]' % (p, n - 2), lev[n - 1])
    for i in range(n - 2, 0, -1):
        lev[i] = rec.s('%s.levIns[%d]' % (p, i), (1 - done[i]) * (1 - isz[i - 1]))
        done[i - 1] = rec.s('%s.done[%d]' % (p, i - 1), lev[i] + done[i])
    lev[0] = rec.s('%s.levIns[0]' % p, 1 - done[0])
    # state machines
    st = []
    prev = dict(top=enabled, i0=0, inew=0, iold=0, na=1 - enabled)
    for i in range(n):
        q = '%s.sm[%d]' % (pfx, i)
        ptl = rec.s(q + '.prev_top_lev_ins', prev['top'] * lev[i])
        ptlf = rec.s(q + '.prev_top_lev_ins_fnc', ptl * fnc)
        cur = dict(top=rec.s(q + '.st_top', prev['top'] - ptl), inew=rec.s(q + '.st_inew', ptl - ptlf),
                   iold=rec.s(q + '.st_iold', 0), i0=rec.s(q + '.st_i0', 0),
                   na=rec.s(q + '.st_na', prev['na'] + prev['inew'] + prev['iold'] + prev['i0']))
        st.append(cur); prev = cur
    # levels, bottom-up
    child = 0
    for i in range(n - 1, -1, -1):
        q = '%s.levels[%d]' % (pfx, i)
        rec.s(q + '.st_top', st[i]['top']); rec.s(q + '.st_inew', st[i]['inew']); rec.s(q + '.st_na', st[i]['na'])
        rec.s(q + '.child', child); rec.s(q + '.sibling', siblings[i]); rec.s(q + '.lrbit', bits[i])
        rec.s(q + '.new1leaf', h1new); rec.s(q + '.switcher.sel', bits[i]); rec.s(q + '.switcher.L', child); rec.s(q + '.switcher.R', siblings[i])
        aux = rec.s(q + '.switcher.aux', (siblings[i] - child) * bits[i])
        L = rec.s(q + '.switcher.outL', aux + child)
        Rr = rec.s(q + '.switcher.outR', siblings[i] - aux)
        rec.s(q + '.proofHash.L', L); rec.s(q + '.proofHash.R', Rr)
        h = poseidon(rec, q + '.proofHash.h', [L, Rr])
        rec.s(q + '.proofHash.out', h)
        a0 = rec.s(q + '.aux[0]', h * st[i]['top'])
        rec.s(q + '.aux[1]', h1old * st[i]['iold'])
        child = rec.s(q + '.root', a0 + h1new * st[i]['inew'])
    # areKeyEquals (oldKey=0 vs key)
    rec.s(pfx + '.areKeyEquals.out', is_zero(rec, pfx + '.areKeyEquals.isz', key))
    is_zero(rec, pfx + '.checkRoot.isz', root - child)
    return child

def census_circuit(inp, nLevels=160):
    """inp: dict of ints / lists like inputs_example.json.  Returns Rec with every template signal."""
    n = nLevels + 1
    rec = Rec()
    g = lambda k: [int(x) % R for x in inp[k]] if isinstance(inp[k], list) else int(inp[k]) % R
    eid, nullifier, aw, vh = g('electionId'), g('nullifier'), g('availableWeight'), g('voteHash')
    sikRoot, censusRoot, address, password, signature, vw = g('sikRoot'), g('censusRoot'), g('address'), g('password'), g('signature'), g('voteWeight')
    cs, ss = g('censusSiblings'), g('sikSiblings')
    rec.s('one', 1)
    for k in ('electionId', 'nullifier', 'availableWeight', 'voteHash', 'sikRoot', 'censusRoot', 'address', 'password', 'signature', 'voteWeight'):
        v = g(k)
        if isinstance(v, list):
            for i, x in enumerate(v): rec.s('main.%s[%d]' % (k, i), x)
        else: rec.s('main.' + k, v)
    for i in range(n): rec.s('main.censusSiblings[%d]' % i, cs[i])
    for i in range(n): rec.s('main.sikSiblings[%d]' % i, ss[i])
    # checkWeight = LessEqThan(252): in0=voteWeight, in1=availableWeight
    x = (vw + (1 << 252) - (aw + 1)) % R
    bits = num2bits(rec, 'main.checkWeight.lt.n2b', x, 253)
    rec.s('main.checkWeight.out', 1 - bits[252])
    sik = poseidon(rec, 'main.sik', [address, password, signature])
    smt_verifier(rec, 'main.sikVerifier', n, sikRoot, ss, address, sik)
    smt_verifier(rec, 'main.censusVerifier', n, censusRoot, cs, address, aw)
    cn = poseidon(rec, 'main.computedNullifier', [signature, password, eid[0], eid[1]])
    is_zero(rec, 'main.checkNullifier.isz', nullifier - cn)
    return rec
