"""CPU (no GPU in this container): the product has no CPU path, and every single-proof entry point says so instead of falling back -- the proving service cannot be
created, groth16_prover returns PROVER_ERROR with the reason, the Node addon rejects fullProve with the same reason and still verifies (verification is host code).
On a GPU box these tests skip: there the same entry points are exercised by the -m gpu suite."""
import ctypes, json, os, shutil, subprocess
import pytest
import oracle_lib as ol


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


pytestmark = pytest.mark.skipif(not _no_gpu(), reason='a GPU is visible: covered by the -m gpu tests')


def test_service_and_rapidsnark_entry_fail_loudly_without_a_gpu():
    from zkcensus_amd import _native, setup, groth16
    lib = _native.load()
    assert not lib.zkc_service_default()
    assert b'no GPU visible' in lib.zkc_service_last_error()
    h = ctypes.c_void_p()
    assert lib.zkc_service_create(None, 0, ctypes.byref(h)) != 0
    _, zkey_path, _ = setup.ensure_test_artifacts(10)
    zk = open(zkey_path, 'rb').read()
    nw = lib.zkc_circuit_n_wires(10)
    payload = (1).to_bytes(32, 'little') + bytes(32 * (nw - 1))
    n = lib.zkc_wtns_write(payload, nw, None, 0); wt = ctypes.create_string_buffer(n); lib.zkc_wtns_write(payload, nw, wt, n)
    ps, us = ctypes.c_ulong(4096), ctypes.c_ulong(4096)
    pb, ub, eb = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096), ctypes.create_string_buffer(256)
    rc = lib.groth16_prover(zk, len(zk), wt.raw, n, pb, ctypes.byref(ps), ub, ctypes.byref(us), eb, 256)
    assert rc == 1 and b'no GPU visible' in eb.value                       # PROVER_ERROR, not a CPU proof
    # the size query and the witness-length check need no GPU and still answer
    ps2, us2 = ctypes.c_ulong(1), ctypes.c_ulong(1)
    assert lib.groth16_prover(zk, len(zk), wt.raw, n, pb, ctypes.byref(ps2), ub, ctypes.byref(us2), eb, 256) == 2 and ps2.value > 600
    short = ctypes.create_string_buffer(lib.zkc_wtns_write(payload, nw - 1, None, 0)); lib.zkc_wtns_write(payload, nw - 1, short, len(short.raw))
    assert lib.groth16_prover(zk, len(zk), short.raw, len(short.raw), pb, ctypes.byref(ps), ub, ctypes.byref(us), eb, 256) == 3


def test_node_addon_rejects_proving_and_still_verifies_without_a_gpu():
    node = shutil.which('node'); addon = os.path.join(ol.ROOT, 'napi', 'zkcensus.node')
    if not node or not os.path.exists(addon):
        pytest.skip('node or the built addon is not available')
    from zkcensus_amd import setup
    _, zkey_path, _ = setup.ensure_test_artifacts(10)
    js = r'''
const { groth16, flatten } = require("./napi/index.js"); const fs = require("fs"); const crypto = require("crypto");
const inp = require("./tests/golden/ref/inputs_example.json");
(async () => {
  const out = { flat: crypto.createHash("sha256").update(flatten(inp, 160)).digest("hex") };
  const vk = JSON.parse(fs.readFileSync("tests/golden/ref/verification_key.json")), pub = JSON.parse(fs.readFileSync("tests/golden/ref/signals.json")), proof = JSON.parse(fs.readFileSync("tests/golden/ref/proof.json"));
  out.verified = await groth16.verify(vk, pub, proof);
  const bad = pub.slice(); bad[0] = String(BigInt(bad[0]) ^ 1n); out.tampered = await groth16.verify(vk, bad, proof);
  try { await groth16.fullProve(Object.assign({}, inp, { censusSiblings: inp.censusSiblings.slice(0, 11), sikSiblings: inp.sikSiblings.slice(0, 11) }), null, process.argv[1], null, { nLevels: 10 }); out.proved = true; }
  catch (e) { out.rejected = String(e); }
  // the addon's 32-byte-word -> decimal-string conversion (what proof.json / public.json are made of) against BigInt, and ragged / numeric / bigint input values through flatten
  const native = require("./napi/zkcensus.node"); let nbad = 0;
  for (let it = 0; it < 500; it++) {
    const b = crypto.randomBytes(64); if (it == 0) b.fill(0); if (it == 1) b.fill(255); if (it == 2) { b.fill(0); b[0] = 1; b[32] = 10; }
    const d = native.decimals(b);
    for (let k = 0; k < 2; k++) if (d[k] !== BigInt("0x" + Buffer.from(b.subarray(32 * k, 32 * k + 32)).reverse().toString("hex")).toString()) nbad++;
  }
  out.decimals_bad = nbad; out.decimals_empty = native.decimals(Buffer.alloc(0)).length;
  const ragged = Object.assign({}, inp, { censusSiblings: inp.censusSiblings.filter((x, i) => i < 12).map((x, i) => i % 2 ? BigInt(x) : x), sikSiblings: inp.sikSiblings.slice(0, 12), voteWeight: Number(inp.voteWeight) });
  out.ragged_equal = Buffer.compare(flatten(ragged, 160), flatten(Object.assign({}, inp, { censusSiblings: inp.censusSiblings.slice(0, 12).concat(Array(149).fill("0")), sikSiblings: inp.sikSiblings.slice(0, 12).concat(Array(149).fill("0")) }), 160)) === 0;
  console.log(JSON.stringify(out));
})();
'''
    r = subprocess.run([node, '-e', js, zkey_path], cwd=ol.ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    import hashlib
    assert j['flat'] == hashlib.sha256(ol.flat_inputs(ol.load_json('ref/inputs_example.json'))).hexdigest()      # the JS and Python flatteners agree byte for byte
    assert j['verified'] is True and j['tampered'] is False                                                    # the reference's own proof triple, through the addon's host path
    assert 'proved' not in j and 'no GPU visible' in j['rejected']
    assert j['decimals_bad'] == 0 and j['decimals_empty'] == 0 and j['ragged_equal'] is True
