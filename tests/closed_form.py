"""TEST INFRASTRUCTURE: the toxic-waste closed form of a Groth16 proof.

The test keys come from a setup with KNOWN toxic waste (csrc/zkc_setup.hip: tau, alpha, beta, gamma, delta drawn from a seed by splitmix64).  Knowing
tau, every group element of a proof is (a field element) x (the generator), and that field element follows from the R1CS and the witness by plain
arithmetic in Fr -- no NTT, no MSM, no .zkey parsing, no H basis, no constant folding:

    a_k = <A_k, w>, b_k = <B_k, w> for every constraint row k (plus snarkjs' rows nCons + i: a = w_i, i <= nPublic), c_k = a_k b_k
    A(tau) = sum a_k L_k(tau), B(tau), C(tau) likewise with the Lagrange basis of the domain AT tau: L_k(tau) = Z(tau) w^k / (n (tau - w^k))
    h(tau) = (A(tau) B(tau) - C(tau)) / Z(tau)                                   [snarkjs groth16_prove.js: the quotient its H section encodes]
    pi_a = (alpha + A(tau) + r delta) G1 ;  pi_b = (beta + B(tau) + s delta) G2
    pi_c = ( [beta A_priv(tau) + alpha B_priv(tau) + C_priv(tau)] / delta + h(tau) Z(tau) / delta + s a + r b - r s delta ) G1
where X_priv sums the private wires only (i > nPublic) and uses the R1CS' own C matrix.  What it shares with the product and with the C oracle
(oracle/groth16.c) is the .r1cs file and the seed; it shares none of their transform, multi-scalar-multiplication or key-file code, so
"GPU proof == closed form" pins exactly the stages that "GPU proof == C oracle proof" leaves to a same-hand restatement.
Python integers throughout: ~1 s at nLevels = 10, ~15 s at nLevels = 160."""
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
G1_GEN = (1).to_bytes(32, 'little') + (2).to_bytes(32, 'little')
G2_GEN = b''.join(x.to_bytes(32, 'little') for x in (
    10857046999023057135944570762232829481370756359578518086990519993285655852781, 11559732032986387107991004021392285783925812861821192530917403151452391805634,
    8495653923123431417604973247489272438418190587263600148770280649306958101930, 4082367875863433681332203403145435568316851327593401208105741076214120093531))


def toxic_waste(seed):
    """csrc/zkc_setup.hip Rng: splitmix64, five elements of 253 random bits each (tau, alpha, beta, gamma, delta)."""
    s = seed & (2**64 - 1); out = []

    def nxt():
        nonlocal s
        s = (s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    for _ in range(5):
        v = 0
        for i in range(4):
            v |= nxt() << (64 * i)
        v &= (1 << 253) - 1                      # word 7 &= 0x1fffffff
        if v & (2**64 - 1) == 0:
            v |= 1
        out.append(v)
    return out


def read_r1cs(path):
    """iden3 .r1cs -> (nWires, nPublic, [(A, B, C)] with each side a list of (wire, coefficient))."""
    buf = open(path, 'rb').read()
    assert buf[:4] == b'r1cs'
    nsec = struct.unpack_from('<I', buf, 8)[0]; p = 12; sec = {}
    for _ in range(nsec):
        sid, n = struct.unpack_from('<IQ', buf, p); p += 12; sec[sid] = (p, n); p += n
    h = sec[1][0]
    assert struct.unpack_from('<I', buf, h)[0] == 32 and int.from_bytes(buf[h + 4:h + 36], 'little') == R
    nWires, nPubOut, nPubIn, _nPrv = struct.unpack_from('<IIII', buf, h + 36); nCons = struct.unpack_from('<I', buf, h + 60)[0]
    q = sec[2][0]; cons = []
    for _ in range(nCons):
        sides = []
        for _m in range(3):
            n = struct.unpack_from('<I', buf, q)[0]; q += 4; terms = []
            for _t in range(n):
                terms.append((struct.unpack_from('<I', buf, q)[0], int.from_bytes(buf[q + 4:q + 36], 'little'))); q += 36
            sides.append(terms)
        cons.append(tuple(sides))
    return nWires, nPubOut + nPubIn, cons


def root_of_unity(logn):
    w = pow(5, (R - 1) >> 28, R)
    for _ in range(28 - logn):
        w = w * w % R
    return w


def proof_scalars(r1cs_path, seed, wtns, r, s):
    """-> (a, b, c): pi_a = a G1, pi_b = b G2, pi_c = c G1 for the witness `wtns` (bytes, nWires x 32 LE, or a list of ints) and blinding (r, s)."""
    nWires, nPub, cons = read_r1cs(r1cs_path)
    w = wtns if isinstance(wtns, list) else [int.from_bytes(wtns[32 * i:32 * i + 32], 'little') for i in range(len(wtns) // 32)]
    assert len(w) == nWires and w[0] == 1
    tau, alpha, beta, gamma, delta = toxic_waste(seed)
    nCons = len(cons)
    logn = 0
    while (1 << logn) < nCons + nPub + 1:
        logn += 1
    n = 1 << logn
    om = root_of_unity(logn)
    Z = (pow(tau, n, R) - 1) % R
    # L_k(tau) for the rows in use, with one batch inversion
    rows = nCons + nPub + 1
    wp = [1] * rows
    for k in range(1, rows):
        wp[k] = wp[k - 1] * om % R
    den = [(tau - x) % R for x in wp]
    pre = [1] * rows; acc = 1
    for k in range(rows):
        pre[k] = acc; acc = acc * den[k] % R
    inv = pow(acc, R - 2, R); L = [0] * rows
    zn = Z * pow(n, R - 2, R) % R
    for k in range(rows - 1, -1, -1):
        L[k] = inv * pre[k] % R * wp[k] % R * zn % R
        inv = inv * den[k] % R
    At = Bt = Ct = Ap = Bp = Cp = 0
    for k, (ra, rb, rc) in enumerate(cons):
        ak = sum(c * w[i] for i, c in ra) % R; bk = sum(c * w[i] for i, c in rb) % R
        At += ak * L[k]; Bt += bk * L[k]; Ct += ak * bk % R * L[k]
        Ap += sum(c * w[i] for i, c in ra if i > nPub) % R * L[k]
        Bp += sum(c * w[i] for i, c in rb if i > nPub) % R * L[k]
        Cp += sum(c * w[i] for i, c in rc if i > nPub) % R * L[k]
        assert ak * bk % R == sum(c * w[i] for i, c in rc) % R, 'constraint %d is not satisfied by this witness' % k
    for i in range(nPub + 1):                     # snarkjs' rows nCons + i (A = wire i, B = 0): they enter A only, and only public wires
        At += w[i] * L[nCons + i]
    At %= R; Bt %= R; Ct %= R
    h = (At * Bt - Ct) % R * pow(Z, R - 2, R) % R
    dinv = pow(delta, R - 2, R)
    a = (alpha + At + r * delta) % R
    b = (beta + Bt + s * delta) % R
    c = ((beta * Ap + alpha * Bp + Cp) % R * dinv + h * Z % R * dinv + s * a + r * b - r * s % R * delta) % R
    return a, b, c


def proof_from_scalars(ol, a, b, c):
    """the 256 proof bytes (A 64 | B 128 | C 64, affine standard form) through the oracle's group arithmetic: one scalar multiplication each"""
    return ol.g1_mul(G1_GEN, a) + ol.msm_g2(G2_GEN, ol.le32(b)) + ol.g1_mul(G1_GEN, c)
