"""CPU: the Node surface's wasm fallback (napi/wasm_witness.js).  groth16.fullProve / wtns.calculate execute a circom-2 witness
calculator this build has no native circuit for, the way snarkjs would (ts_inputs/src/example.ts:358-362 hands snarkjs a wasm path).  No GPU here: wtns.calculate only --
the proving half runs on the GPU box (tests/test_00_gpu_node_addon.py)."""
import hashlib, json, os, shutil, subprocess, sys
import pytest
import oracle_lib as ol

REF_WASM = '/root/reference/artifacts/zkCensus/dev/160/circuit.wasm'


def _node(js, *argv):
    node = shutil.which('node'); addon = os.path.join(ol.ROOT, 'napi', 'zkcensus.node')
    if not node or not os.path.exists(addon):
        pytest.skip('node or the built addon is not available')
    r = subprocess.run([node, '-e', js] + list(argv), cwd=ol.ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_toy_wasm_is_what_the_assembler_writes():
    sys.path.insert(0, os.path.join(ol.ROOT, 'tools'))
    import make_toy_wasm
    assert open(ol.golden('toy_passthrough.wasm'), 'rb').read() == make_toy_wasm.build()


def test_unknown_wasm_is_executed_and_yields_a_wtns_image():
    js = r'''
const { wtns } = require("./napi/index.js");
(async () => {
  const W = "tests/golden/toy_passthrough.wasm", out = {};
  const img = await wtns.calculate({ a: "5", b: ["7"] }, W, { type: "mem" });
  out.header = img.subarray(0, 4).toString("latin1") + ":" + img.readUInt32LE(4) + ":" + img.readUInt32LE(8) + ":" + img.readUInt32LE(24 + 36);
  out.words = [0, 1, 2, 3].map((i) => img.subarray(76 + 32 * i, 108 + 32 * i).toString("hex"));
  const big = await wtns.calculate({ b: "21888242871839275222246405745257275088548364400416034343698204186575808495618", a: "-1" }, W, { type: "mem" });
  out.reduced = [1, 3].map((i) => big.subarray(76 + 32 * i, 108 + 32 * i).toString("hex"));
  out.errors = [];
  for (const bad of [{ a: "0", b: "1" }, { a: "1", c: "1" }, { a: "1" }, { a: ["1", "2"], b: "1" }]) {
    try { await wtns.calculate(bad, W, { type: "mem" }); out.errors.push(null); } catch (e) { out.errors.push(e.message); }
  }
  const again = await wtns.calculate({ a: "9", b: "1" }, W, { type: "mem" });          // after four failed runs the calculator still works
  out.again = again.readUInt32LE(76 + 32);
  try { await wtns.calculate({ a: "1", b: "2" }, Buffer.from("not a circuit"), { type: "mem" }); } catch (e) { out.notWasm = e.message; }
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
'''
    j = _node(js)
    le = lambda x: x.to_bytes(32, 'little').hex()
    assert j['header'] == 'wtns:2:2:4'                                   # magic, version 2, two sections, four witnesses
    assert j['words'] == [le(1), le(5), le(5), le(7)]
    assert j['reduced'] == [le(ol.R - 1), le(1)]                          # inputs are taken mod r, negative ones included (circom_runtime does the same)
    assert j['errors'][0] == 'Assert Failed.\nError in template Toy_0 line: 7\n'      # code text + the wasm's own message: the shape of snarkjs' Error.message
    assert j['errors'][1] == 'Signal not found.\n' and 'Not all inputs have been set' in j['errors'][2] and 'Too many values for input signal a' in j['errors'][3]
    assert j['again'] == 9
    assert 'unknown circuit wasm' in j['notWasm'] and 'cannot be executed' in j['notWasm']


@pytest.mark.skipif(not os.path.exists(REF_WASM), reason='the reference tree exists in the build container only')
def test_fallback_reproduces_the_reference_wasm_witness():
    """forceWasm on the reference's own dev/160 circuit.wasm: the witness the Node fallback computes is the one SURVEY.md A.4 fingerprints (and the native generator reproduces),
    and a violated assert carries the reference's message"""
    js = r'''
const { wtns } = require("./napi/index.js");
const inputs = require("./tests/golden/ref/inputs_example.json");
(async () => {
  const W = process.argv[1], out = {};
  const img = await wtns.calculate(inputs, W, { type: "mem" }, { forceWasm: true });
  out.sha = require("crypto").createHash("sha256").update(img.subarray(76)).digest("hex");
  try { await wtns.calculate(Object.assign({}, inputs, { nullifier: "1" }), W, { type: "mem" }, { forceWasm: true }); } catch (e) { out.err = e.message; }
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
'''
    j = _node(js, REF_WASM)
    assert j['sha'] == 'ebf5467e953a0427fa50c9a0b0521ac1c5c70684ef3603c75807c1bc4315e71b'
    assert j['err'] == 'Assert Failed.\nError in template ForceEqualIfEnabled_159 line: 56\nError in template ZkFranchiseProofCircuit_234 line: 114\n'
