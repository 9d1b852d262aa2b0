"""Synthetic census / voter generator: arbo-compatible Poseidon sparse Merkle trees with sibling
extraction, SIK and nullifier derivation.  Mirrors internal/helpers.go:36-85 (GenTree: arbo.NewTree{Poseidon}, Add,
GenProof, UnpackSiblings, zero padding) and internal/inputs.go:33-98 (MockInputs); encodings per
ts_inputs/src/inputs.ts:55-88.  All hashing runs on the GPU through zkc_poseidon_batch, one call per tree level.

arbo tree semantics: leaf = H(key, value, 1); node = H(left, right); path bit i = bit i (LSB first) of
the key; an empty subtree is 0; a subtree holding a single leaf is that leaf's hash (the leaf sits at the first level
where its path is unique)."""
import ctypes
import hashlib
from .inputs import R_MOD, bytes_to_arbo  # noqa: F401 (bytes_to_arbo is part of this module's surface)

ELECTION_ID_HEX = '7faeab7a7d250527d614e952ae8e446825bd1124c6def410844c7c383d1519a6'   # internal/inputs.go:60, example.ts:341


def poseidon_batch(ctx, rows):
    """rows: list of equal-length tuples of ints (2, 3 or 4 inputs) -> list of ints, hashed on the GPU."""
    if not rows:
        return []
    n = len(rows[0])
    buf = b''.join(int(x).to_bytes(32, 'little') for r in rows for x in r)
    out = ctypes.create_string_buffer(32 * len(rows))
    ctx._check(ctx._lib.zkc_poseidon_batch(ctx._h, n, buf, len(rows), out))
    raw = out.raw
    return [int.from_bytes(raw[32 * i:32 * i + 32], 'little') for i in range(len(rows))]


class SparseMerkleTree:
    """Static build over a list of (key, value) pairs; records for every leaf its sibling path."""

    def __init__(self, ctx, keys, values, max_levels=160):
        assert len(set(keys)) == len(keys), 'duplicate keys'
        self.n = len(keys)
        leaf_hash = poseidon_batch(ctx, [(k, v, 1) for k, v in zip(keys, values)])
        # top-down split into the radix trie; node = (depth, members) ; children by key bit `depth`
        nodes = []            # (depth, left_id, right_id, members)   id -1 = empty, -(2+i) = single leaf i
        def ref(members, depth):
            if not members:
                return -1
            if len(members) == 1:
                return -(2 + members[0])
            if depth >= max_levels:
                raise ValueError('keys collide on the first %d bits' % max_levels)
            nid = len(nodes); nodes.append(None)
            l = [m for m in members if not (keys[m] >> depth) & 1]; r = [m for m in members if (keys[m] >> depth) & 1]
            nodes[nid] = [depth, None, None, members]
            pending.append((nid, l, r, depth))
            return nid
        pending = []
        import sys
        root_ref = ref(list(range(self.n)), 0)
        while pending:                                   # iterative expansion (no recursion depth issues)
            nid, l, r, depth = pending.pop()
            nodes[nid][1] = ref(l, depth + 1); nodes[nid][2] = ref(r, depth + 1)
        # bottom-up hashing, one GPU batch per depth
        val = {}
        def value_of(rid):
            return 0 if rid == -1 else leaf_hash[-rid - 2] if rid < -1 else val[rid]
        by_depth = {}
        for nid, nd in enumerate(nodes):
            by_depth.setdefault(nd[0], []).append(nid)
        for depth in sorted(by_depth, reverse=True):
            ids = by_depth[depth]
            hs = poseidon_batch(ctx, [(value_of(nodes[i][1]), value_of(nodes[i][2])) for i in ids])
            for i, h in zip(ids, hs):
                val[i] = h
        self.root = value_of(root_ref)
        # sibling paths
        self.siblings = [[] for _ in range(self.n)]
        for nd in nodes:
            depth, lref, rref, members = nd
            lv, rv = value_of(lref), value_of(rref)
            for m in members:
                sib = lv if (keys[m] >> depth) & 1 else rv
                path = self.siblings[m]
                while len(path) < depth:
                    path.append(0)
                if len(path) == depth:
                    path.append(sib)
                else:
                    path[depth] = sib
        for p in self.siblings:                          # drop trailing zeros (GenProof packs only used levels)
            while p and p[-1] == 0:
                p.pop()


def _voter_data(n_voters, election_id_hex):
    """the deterministic raw data of the synthetic census (addresses, passwords, signatures, weights)."""
    eid = bytes_to_arbo(bytes.fromhex(election_id_hex))
    u32 = lambda i: int(i).to_bytes(4, 'little')
    address = [int.from_bytes(hashlib.sha256(b'addr' + u32(i)).digest()[:20], 'little') for i in range(n_voters)]
    password = [int.from_bytes(hashlib.sha256(b'pw' + u32(i)).digest()[:11], 'big') % R_MOD for i in range(n_voters)]
    signature = [int.from_bytes(hashlib.sha256(b'sigA' + u32(i)).digest() + hashlib.sha256(b'sigB' + u32(i)).digest(), 'big') % R_MOD
                 for i in range(n_voters)]
    avail = [1 + (i % 100) for i in range(n_voters)]
    return eid, address, password, signature, avail


def census_inputs(ctx, election_id, address, password, signature, available_weight, vote_weight, vote_hash, nLevels=160, d_out_ptr=None):
    """[r5] The native census builder (zkc_census_inputs, csrc/zkc_census.hip): every voter's circuit inputs -- SIK, nullifier, both trees, roots, sibling lists -- as flat
    334 x 32-byte blocks, all hashing and the sibling scatter on the GPU.  Lists of ints (vote_hash: pairs).  Returns (flat bytes, census root, sik root); with d_out_ptr
    the blocks are ALSO left at that device address."""
    n = len(address)
    le = lambda xs: b''.join(int(x).to_bytes(32, 'little') for x in xs)
    nIn = 12 + 2 * (nLevels + 1)
    out = ctypes.create_string_buffer(32 * nIn * n); roots = ctypes.create_string_buffer(64)
    ctx._check(ctx._lib.zkc_census_inputs(ctx._h, n, nLevels, le(election_id), le(address), le(password), le(signature), le(available_weight), le(vote_weight),
                                          le(x for pair in vote_hash for x in pair), ctypes.cast(out, ctypes.c_void_p), d_out_ptr, roots))
    return out.raw, int.from_bytes(roots.raw[:32], 'little'), int.from_bytes(roots.raw[32:], 'little')


def synthetic_census_flat(ctx, n_voters, nLevels=160, election_id_hex=ELECTION_ID_HEX, d_out_ptr=None):
    """The synthetic census of SURVEY.md 8(d) config 3 through the native builder: (flat input blocks, census root, sik root).  8 192 voters in well under a second
    (the Python builder below, kept as a cross-check, takes ten)."""
    eid, address, password, signature, avail = _voter_data(n_voters, election_id_hex)
    vh = [bytes_to_arbo(a.to_bytes((a.bit_length() + 7) // 8 or 1, 'big')) for a in avail]                  # internal/inputs.go:81
    return census_inputs(ctx, eid, address, password, signature, avail, [1] * n_voters, vh, nLevels, d_out_ptr)


class FlatVoters:
    """The 12-key input objects of a flat block array, made on demand (a decimal string per value costs more than the GPU spends on the voter's proof: only the voters
    somebody looks at are converted)."""

    def __init__(self, flat, nLevels=160):
        self.flat, self.nLevels, self.nIn = flat, nLevels, 12 + 2 * (nLevels + 1)

    def __len__(self):
        return len(self.flat) // (32 * self.nIn)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        b = self.flat[32 * self.nIn * i:32 * self.nIn * (i + 1)]
        v = [str(int.from_bytes(b[32 * k:32 * k + 32], 'little')) for k in range(self.nIn)]
        n = self.nLevels + 1
        return {'electionId': v[0:2], 'nullifier': v[2], 'availableWeight': v[3], 'voteHash': v[4:6], 'sikRoot': v[6], 'censusRoot': v[7], 'address': v[8], 'password': v[9],
                'signature': v[10], 'voteWeight': v[11], 'censusSiblings': v[12:12 + n], 'sikSiblings': v[12 + n:12 + 2 * n]}


def synthetic_census(ctx, n_voters, nLevels=160, election_id_hex=ELECTION_ID_HEX):
    """deterministic census of n_voters as a sequence of 12-key circuit input objects (decimal strings, siblings zero-padded to nLevels + 1 like
    internal/inputs.go:52,72).  [r5] Built by the native builder (zkc_census_inputs); the objects are made on demand from its flat blocks (FlatVoters: indexing and slicing give
    dicts / lists of dicts, `.flat` the blocks themselves).  synthetic_census_py is the Python builder of rounds 1-4, kept as the cross-check
    (tests/test_gpu_census.py: byte-equal over whole censuses)."""
    flat, _, _ = synthetic_census_flat(ctx, n_voters, nLevels, election_id_hex)
    return FlatVoters(flat, nLevels)


def synthetic_census_py(ctx, n_voters, nLevels=160, election_id_hex=ELECTION_ID_HEX):
    """The same census through SparseMerkleTree above (Python lists, one batched GPU Poseidon call per tree level): ten seconds for 8 192 voters."""
    eid, address, password, signature, avail = _voter_data(n_voters, election_id_hex)
    sik = poseidon_batch(ctx, list(zip(address, password, signature)))                     # census.circom:74-77
    nullifier = poseidon_batch(ctx, [(s, p, int(eid[0]), int(eid[1])) for s, p in zip(signature, password)])   # :105-109
    census = SparseMerkleTree(ctx, address, avail, nLevels)
    siktree = SparseMerkleTree(ctx, address, sik, nLevels)
    pad = lambda s: [str(x) for x in s] + ['0'] * (nLevels + 1 - len(s))
    out = []
    for i in range(n_voters):
        vh = bytes_to_arbo(avail[i].to_bytes((avail[i].bit_length() + 7) // 8 or 1, 'big'))   # internal/inputs.go:81
        out.append({
            'electionId': list(eid), 'nullifier': str(nullifier[i]), 'availableWeight': str(avail[i]), 'voteHash': vh,
            'sikRoot': str(siktree.root), 'censusRoot': str(census.root),
            'address': str(address[i]), 'password': str(password[i]), 'signature': str(signature[i]), 'voteWeight': '1',
            'censusSiblings': pad(census.siblings[i]), 'sikSiblings': pad(siktree.siblings[i]),
        })
    return out


def deep_voters(ctx, n_voters, nLevels=160, depth=None, seed=160):
    """n_voters valid voters whose leaves sit `depth` levels down BOTH trees (default nLevels: the bottom), every sibling on the way non-zero: no level of
    their witnesses equals the voter-independent empty-subtree template, so the prover's constant folding removes nothing -- the worst case a real census
    cannot produce (8 192 voters put leaves 13-17 levels deep) but a foreign or adversarial witness can.  Each voter gets its own pair of roots: the
    proofs are independent anyway.  Hashing on the GPU, one zkc_poseidon_batch call per level."""
    import random
    rng = random.Random(seed); depth = nLevels if depth is None else depth
    eid = bytes_to_arbo(bytes.fromhex(ELECTION_ID_HEX))
    address = [rng.getrandbits(160) for _ in range(n_voters)]
    password = [rng.getrandbits(88) for _ in range(n_voters)]
    signature = [rng.getrandbits(512) % R_MOD for _ in range(n_voters)]
    avail = [1 + rng.randrange(100) for _ in range(n_voters)]
    sik = poseidon_batch(ctx, list(zip(address, password, signature)))
    nullifier = poseidon_batch(ctx, [(s, p, int(eid[0]), int(eid[1])) for s, p in zip(signature, password)])

    def climb(values):
        sibs = [[rng.randrange(1, R_MOD) for _ in range(depth)] for _ in range(n_voters)]
        cur = poseidon_batch(ctx, [(k, v, 1) for k, v in zip(address, values)])
        for lvl in range(depth - 1, -1, -1):
            cur = poseidon_batch(ctx, [(sibs[v][lvl], cur[v]) if (address[v] >> lvl) & 1 else (cur[v], sibs[v][lvl]) for v in range(n_voters)])
        return cur, sibs
    croot, csib = climb(avail); sroot, ssib = climb(sik)
    pad = lambda s: [str(x) for x in s] + ['0'] * (nLevels + 1 - len(s))
    return [{'electionId': list(eid), 'nullifier': str(nullifier[i]), 'availableWeight': str(avail[i]), 'voteHash': ['1', '2'], 'sikRoot': str(sroot[i]),
             'censusRoot': str(croot[i]), 'address': str(address[i]), 'password': str(password[i]), 'signature': str(signature[i]), 'voteWeight': '1',
             'censusSiblings': pad(csib[i]), 'sikSiblings': pad(ssib[i])} for i in range(n_voters)]
