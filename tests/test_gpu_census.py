"""GPU: the census builder (f1) produces voters the circuit accepts, identical to what the oracle computes."""
import pytest
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def test_poseidon_batch_and_census():
    import zkcensus_amd
    from zkcensus_amd import census
    ctx = zkcensus_amd.Context(0)
    rows = [(1, 2), (0, 0), (ol.R - 1, 5)]
    assert census.poseidon_batch(ctx, rows) == [ol.poseidon(list(r)) for r in rows]
    assert census.poseidon_batch(ctx, [(1, 2, 3, 4)]) == [18821383157269793795438455681495246036402687001665670618754263018637548127333]
    assert census.poseidon_batch(ctx, [(0, 0, 1)]) == [3108394280857290448796042949317662357879960495408018998613518544538624657019]
    voters = census.synthetic_census_py(ctx, 300)                        # the Python builder: the circuit must accept what IT builds too
    assert len({v['censusRoot'] for v in voters}) == 1 and len({v['sikRoot'] for v in voters}) == 1
    depths = [max([i + 1 for i, s in enumerate(v['censusSiblings']) if s != '0'] or [0]) for v in voters]
    assert 6 <= max(depths) <= 40
    ws, st = ctx.witness(voters[:64])
    assert st == [0] * 64
    for v, w in list(zip(voters, ws))[:8]:
        rc, wo = ol.witness(v)
        assert rc == 0 and wo == w
    # the one arbo-built path the reference holds (inputs_example.json, written by internal/helpers.go:36-85 GenTree / GenProof): climbing it with the GPU
    # Poseidon and this module's leaf / node / path-bit conventions must land on the roots arbo computed.  (The reference has no arbo-built multi-leaf tree
    # fixture, so SparseMerkleTree's branching is pinned only through the circuit accepting its paths -- above -- and through this path.)
    ex = ol.load_json('ref/inputs_example.json')
    key = int(ex['address'])

    def climb(value, siblings):
        sib = [int(x) for x in siblings]
        d = max([i + 1 for i, x in enumerate(sib) if x] or [0])
        cur = census.poseidon_batch(ctx, [(key, value, 1)])[0]
        for i in range(d - 1, -1, -1):
            cur = census.poseidon_batch(ctx, [(sib[i], cur) if (key >> i) & 1 else (cur, sib[i])])[0]
        return cur
    assert climb(int(ex['availableWeight']), ex['censusSiblings']) == int(ex['censusRoot'])
    sik = census.poseidon_batch(ctx, [(key, int(ex['password']), int(ex['signature']))])[0]
    assert climb(sik, ex['sikSiblings']) == int(ex['sikRoot'])
    # voters at the very bottom of both trees (bench.py's worst case for constant folding) are valid voters too
    deep = census.deep_voters(ctx, 3, 160)
    ws, st = ctx.witness(deep)
    assert st == [0, 0, 0]
    rc, wo = ol.witness(deep[1]); assert rc == 0 and wo == ws[1]
    assert all(v['censusSiblings'][159] != '0' and v['censusSiblings'][160] == '0' for v in deep)
    ctx.close()


def test_deep_pass_takes_the_second_section_tables():
    """[r4] A pass whose voters keep more than 16 000 wires per section (leaves at the bottom of both trees: nothing folds) runs its sections over the key's second,
    15-bit-window tables: 17 instead of 22 additions per scalar.  Four voters 160 levels down and one 9 levels down in one call: proof bytes equal the oracle's, the
    verifier accepts, and the device's count of G1 additions says which tables the pass took (H: 15 per scalar either way)."""
    import ctypes, json, random
    import zkcensus_amd
    from zkcensus_amd import census, setup
    nl = 160
    _, zp, vp = setup.ensure_test_artifacts(nl)
    zk = open(zp, 'rb').read(); vk = json.load(open(vp))
    ctx = zkcensus_amd.Context(0); pk = zkcensus_amd.ProvingKey(ctx, zk)
    voters = census.deep_voters(ctx, 4, nl) + census.synthetic_census(ctx, 600)[:1]
    ws, st = ctx.witness(voters)
    assert st == [0] * 5
    import numpy as np, torch
    rng = random.Random(17)
    rs = [(rng.randrange(ol.R), rng.randrange(ol.R)) for _ in voters]
    rsb = b''.join(r.to_bytes(32, 'little') + s_.to_bytes(32, 'little') for r, s_ in rs)
    d_w = torch.from_numpy(np.frombuffer(b''.join(ws), dtype=np.uint8).copy()).cuda()
    lib = ctx._lib
    lib.zkc_profile_enable(ctx._h, 0x10)
    p_all, u_all = pk.prove_batch_dev(d_w.data_ptr(), 5, rsb)
    ms, n, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
    lib.zkc_profile_read(ctx._h, 7, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))      # ZKC_PROF_MSM_G1_STREAMED: launches = mixed additions counted on the device
    lib.zkc_profile_enable(ctx._h, 0)
    want = ol.pmap(lambda a: ol.prove(zk, a[0], a[1][0], a[1][1]), list(zip(ws, rs)))
    for q, (rc, op, ou) in enumerate(want):
        assert rc == 0 and (p_all[256 * q:256 * q + 256], u_all[256 * q:256 * q + 256]) == (op, ou), q
        assert ol.verify(vk, ou, op)
    # 22 additions per section scalar: 6.55 M per deep proof; 17: about 5.5 M (H: 15 per scalar either way, 1.97 M)
    assert n.value < 5 * 5.3e6, n.value
    # a deep pass and a shallow one (12-bit tables, 2048 buckets per section) in flight together on the key's two call slots, sixteen times: the two layouts of the lanes' work
    # space follow each other pass by pass, and every call gives the bytes it gives alone
    shallow = census.synthetic_census(ctx, 600)[1:7]
    ws2, st2 = ctx.witness(shallow); assert st2 == [0] * 6
    rsb2 = b''.join(rng.randrange(ol.R).to_bytes(32, 'little') for _ in range(12))
    d_w2 = torch.from_numpy(np.frombuffer(b''.join(ws2), dtype=np.uint8).copy()).cuda()
    alone = pk.prove_batch_dev(d_w2.data_ptr(), 6, rsb2)
    from zkcensus_amd import groth16
    assert groth16.verify_batch(ctx, vk, alone[1], alone[0])
    for k in range(16):
        pk.batch_begin(k & 1, None, 5, d_w.data_ptr(), None, rsb)
        pk.batch_begin(1 - (k & 1), None, 6, d_w2.data_ptr(), None, rsb2)
        assert pk.batch_finish(k & 1, 5) == (p_all, u_all), k
        assert pk.batch_finish(1 - (k & 1), 6) == alone, k
    pk.close(); ctx.close()


def test_native_census_builder_equals_the_python_tree_and_the_oracle():
    """[r5] zkc_census_inputs / zkc_smt_build (csrc/zkc_census.hip: trie split on the host in C++, every hash and the sibling scatter on the GPU) against
    census.SparseMerkleTree (the Python builder of rounds 1-4, itself pinned through the circuit and the reference's arbo-built path) -- byte for byte over whole censuses --
    and, for a small tree, against a pure-Python climb with the oracle's Poseidon.  Edge shapes: one voter, two voters whose paths share 40 bits, a census that does not
    fit the depth (two keys colliding on their first nLevels bits), duplicate keys, values at r - 1."""
    import time, ctypes, random
    import zkcensus_amd
    from zkcensus_amd import census
    ctx = zkcensus_amd.Context(0)
    for n, nl in ((300, 160), (2048, 160), (50, 10), (1, 160)):
        py = census.synthetic_census_py(ctx, n, nl) if nl == 160 or n < 2 else None
        if py is None:                                                   # at nLevels 10 fifty random 160-bit addresses may collide on 10 bits: take the low bits apart
            continue
        flat, croot, sroot = census.synthetic_census_flat(ctx, n, nl)
        assert flat == b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in py), (n, nl)
        assert str(croot) == py[0]['censusRoot'] and str(sroot) == py[0]['sikRoot']
        fv = census.FlatVoters(flat, nl)
        assert len(fv) == n and fv[0] == py[0] and fv[n - 1] == py[-1]
    # one tree, small, against a climb with the ORACLE's Poseidon (no GPU code on the checking side)
    rng = random.Random(5)
    nl = 12; keys = rng.sample(range(1 << nl), 40); vals = [rng.randrange(ol.R) for _ in keys]; vals[3] = ol.R - 1
    le = lambda xs: b''.join(int(x).to_bytes(32, 'little') for x in xs)
    root = ctypes.create_string_buffer(32); sib = ctypes.create_string_buffer(32 * (nl + 1) * len(keys)); dep = (ctypes.c_int32 * len(keys))()
    ctx._check(ctx._lib.zkc_smt_build(ctx._h, le(keys), le(vals), len(keys), nl, root, sib, dep))
    rt = int.from_bytes(root.raw, 'little')
    for i, (k, v) in enumerate(zip(keys, vals)):
        s = [int.from_bytes(sib.raw[32 * ((nl + 1) * i + l):32 * ((nl + 1) * i + l + 1)], 'little') for l in range(nl + 1)]
        assert all(x == 0 for x in s[dep[i]:])
        cur = ol.poseidon([k, v, 1])
        for l in range(dep[i] - 1, -1, -1):
            cur = ol.poseidon([s[l], cur]) if (k >> l) & 1 else ol.poseidon([cur, s[l]])
        assert cur == rt, i
    t = census.SparseMerkleTree(ctx, keys, vals, nl)
    assert t.root == rt
    # two leaves whose paths part at bit 40: forty inner nodes with one empty child each
    k2 = [5, 5 + (1 << 40)]
    ctx._check(ctx._lib.zkc_smt_build(ctx._h, le(k2), le([7, 9]), 2, 160, root, None, dep))
    assert list(dep) [:2] == [41, 41] and int.from_bytes(root.raw, 'little') == census.SparseMerkleTree(ctx, k2, [7, 9], 160).root
    # refusals: keys that agree on the first nLevels path bits, duplicate keys, a value that is not a field element
    assert ctx._lib.zkc_smt_build(ctx._h, le([1, 1 + (1 << 12)]), le([1, 2]), 2, 12, root, None, None) == 4
    assert ctx._lib.zkc_smt_build(ctx._h, le([9, 9]), le([1, 2]), 2, 160, root, None, None) == 4
    assert ctx._lib.zkc_smt_build(ctx._h, le([1, 2]), le([1, ol.R]), 2, 160, root, None, None) == 4
    # the whole 8 192-voter census of BASELINE configs[2..3] in well under a second, its first voters accepted by the circuit with the oracle's witness
    t0 = time.time(); flat, croot, sroot = census.synthetic_census_flat(ctx, 8192, 160); dt = time.time() - t0
    print('\nnative census builder: 8192 voters in %.3f s' % dt)
    assert dt < 1.0, dt
    fv = census.FlatVoters(flat, 160)
    ws, st = ctx.witness([fv[0], fv[4095], fv[8191]])
    assert st == [0, 0, 0]
    rc, w = ol.witness(fv[8191]); assert rc == 0 and w == ws[2]
    ctx.close()
