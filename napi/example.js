// The reference's ts_inputs/src/example.ts:358-362 call, against this package (needs an MI355X and a test zkey):
//     node napi/example.js <zkey> [verification_key.json] [circuit.wasm | -] [voters.json | -] [toy.zkey toy_vkey.json [nl10.zkey nl10_vkey.json nl10_voter.json]]
// With a wasm path the circuit is selected by its sha256 exactly as a snarkjs caller names it; without one the native nLevels = 160
// circuit is used.  Also drives the two-step path (wtns.calculate -> groth16.prove), four concurrent fullProve calls and, with a file of voters
// (a JSON array of input objects), the reference's call shape under load: Promise.all over one fullProve PER VOTER, which the library's proving service
// coalesces into pipeline passes.
const { groth16, wtns } = require("./index.js");
const fs = require("fs");
const inputs = require("../tests/golden/ref/inputs_example.json");
(async () => {
  const zkey = process.argv[2], vk = process.argv[3] ? JSON.parse(fs.readFileSync(process.argv[3])) : null;
  const wasm = process.argv[4] && process.argv[4] !== "-" && fs.existsSync(process.argv[4]) ? process.argv[4] : null;
  const t0 = Date.now();
  const { proof, publicSignals } = await groth16.fullProve(inputs, wasm, zkey);
  const ms = Date.now() - t0;
  const verified = vk ? await groth16.verify(vk, publicSignals, proof) : null;
  let msWarm = 1e9;                                  // key resident, artifact cached: what a service sees per voter
  for (let i = 0; i < 5; i++) { const t = process.hrtime.bigint(); await groth16.fullProve(inputs, wasm, zkey); msWarm = Math.min(msWarm, Number(process.hrtime.bigint() - t) / 1e6); }
  // two-step path with injected (r, s): deterministic, equal to fullProve with the same scalars
  const mem = { type: "mem" };
  await wtns.calculate(inputs, wasm, mem);
  const a = await groth16.prove(zkey, mem, null, { r: 12345n, s: 67890n });
  const b = await groth16.fullProve(inputs, wasm, zkey, null, { r: 12345n, s: 67890n });
  const twoStepEqual = JSON.stringify(a) === JSON.stringify(b) && (!vk || await groth16.verify(vk, a.publicSignals, a.proof));
  // submitted back to back from this thread: the proving service takes them in one or two passes
  const many = await Promise.all([0, 1, 2, 3].map(() => groth16.fullProve(inputs, wasm, zkey)));
  let concurrentOk = true;
  for (const m of many) concurrentOk = concurrentOk && JSON.stringify(m.publicSignals) === JSON.stringify(publicSignals) && (!vk || await groth16.verify(vk, m.publicSignals, m.proof));
  // one fullProve per voter, all at once (what a ballot-box service does with example.ts:358): every proof verified, rate = voters / wall time
  let burst = null;
  if (process.argv[5] && process.argv[5] !== "-") {
    const voters = JSON.parse(fs.readFileSync(process.argv[5]));
    const run = async (list) => { const t = process.hrtime.bigint(); const out = await Promise.all(list.map((v) => groth16.fullProve(v, wasm, zkey))); return [out, Number(process.hrtime.bigint() - t) / 1e6]; };
    const many4 = [].concat(voters, voters, voters, voters);
    await run(voters); await run(many4);                                 // the library's work space grows to a full pass once (first burst only)
    const [out64, ms64] = await run(voters);
    const [out256, ms256] = await run(many4);
    let allVerified = true, signalsOk = true;
    for (let i = 0; i < out64.length; i++) {
      signalsOk = signalsOk && out64[i].publicSignals[2] === String(voters[i].nullifier) && out64[i].publicSignals[6] === String(voters[i].censusRoot);
      if (vk) allVerified = allVerified && await groth16.verify(vk, out64[i].publicSignals, out64[i].proof);
    }
    for (let i = 0; i < out256.length; i += 5) if (vk) allVerified = allVerified && await groth16.verify(vk, out256[i].publicSignals, out256[i].proof);
    burst = { voters: voters.length, ms: Math.round(ms64 * 10) / 10, proofsPerSec: Math.round(voters.length / ms64 * 1e3), voters4x: many4.length, ms4x: Math.round(ms256 * 10) / 10,
      proofsPerSec4x: Math.round(many4.length / ms256 * 1e3), allVerified, signalsOk };
  }
  // the message of the Error snarkjs throws for this voter with the reference's circuit.wasm (tests/golden/witness_vectors.json "bad_nullifier")
  const NULLIFIER_ASSERT = "Assert Failed.\nError in template ForceEqualIfEnabled_159 line: 56\nError in template ZkFranchiseProofCircuit_234 line: 114\n";
  // a batch over a pool of devices (device 0 listed twice: two contexts, two host threads): voter 1 fails an assert, the others equal fullProve with the same (r, s)
  const batch = await groth16.fullProveBatch([inputs, Object.assign({}, inputs, { nullifier: "1" }), inputs], wasm, zkey,
    { devices: [0, 0], rs: [[12345n, 67890n], [1n, 2n], [12345n, 67890n]] });
  const batchOk = batch.length === 3 && JSON.stringify(batch[0]) === JSON.stringify(b) && JSON.stringify(batch[2]) === JSON.stringify(b) &&
    batch[1] instanceof Error && batch[1].message === NULLIFIER_ASSERT;
  let badInputRejected = false;
  try { await groth16.fullProve(Object.assign({}, inputs, { nullifier: "1" }), wasm, zkey); } catch (e) { badInputRejected = e.message === NULLIFIER_ASSERT; }
  let unknownWasmRejected = false;
  try { await groth16.fullProve(inputs, Buffer.from("not a circuit"), zkey); } catch (e) { unknownWasmRejected = /unknown circuit wasm/.test(String(e)); }
  // [r4] a circuit this build has NO native witness generator for: the caller's wasm is executed in Node (napi/wasm_witness.js), the proof is made on the GPU from that
  // witness.  argv[6..7] = key and verification key of tests/golden/toy_passthrough.wasm's circuit (out <== a; assert(a != 0)); argv[8..10] = an nLevels-10 census key, its
  // verification key and a voter for it: wasmFile null, no opts.nLevels -- the depth is read off the key.
  let wasmFallback = null, depthFromKey = null;
  if (process.argv[7]) {
    const toyWasm = require("path").join(__dirname, "..", "tests", "golden", "toy_passthrough.wasm"), toyZkey = process.argv[6], toyVk = JSON.parse(fs.readFileSync(process.argv[7]));
    const t = await groth16.fullProve({ a: "5", b: "7" }, toyWasm, toyZkey);
    const proves = t.publicSignals.length === 1 && t.publicSignals[0] === "5" && await groth16.verify(toyVk, t.publicSignals, t.proof);
    const m2 = { type: "mem" };
    await wtns.calculate({ a: "5", b: "7" }, toyWasm, m2);
    const p1 = await groth16.prove(toyZkey, m2, null, { r: 3n, s: 4n }), p2 = await groth16.fullProve({ a: "5", b: "7" }, toyWasm, toyZkey, null, { r: 3n, s: 4n });
    const twoStep = JSON.stringify(p1) === JSON.stringify(p2);
    let assertText = false;
    try { await groth16.fullProve({ a: "0", b: "7" }, toyWasm, toyZkey); } catch (e) { assertText = e.message === "Assert Failed.\nError in template Toy_0 line: 7\n"; }
    const many16 = await Promise.all([...Array(16).keys()].map((i) => groth16.fullProve({ a: String(i + 1), b: "9" }, toyWasm, toyZkey)));
    let burstOk = true;
    for (let i = 0; i < 16; i++) burstOk = burstOk && many16[i].publicSignals[0] === String(i + 1) && await groth16.verify(toyVk, many16[i].publicSignals, many16[i].proof);
    const bt = await groth16.fullProveBatch([{ a: "5", b: "7" }, { a: "0", b: "1" }], toyWasm, toyZkey, { rs: [[3n, 4n], [1n, 2n]] });
    const batchOkW = JSON.stringify(bt[0]) === JSON.stringify(p2) && bt[1] instanceof Error && /Assert Failed/.test(bt[1].message);
    let noWasmRefused = false;
    try { await groth16.fullProve({ a: "1", b: "2" }, null, toyZkey); } catch (e) { noWasmRefused = /not a ZkFranchiseProofCircuit key/.test(e.message); }
    wasmFallback = { proves, twoStep, assertText, burstOk, batchOk: batchOkW, noWasmRefused };
  }
  if (process.argv[10]) {
    const vk10 = JSON.parse(fs.readFileSync(process.argv[9])), v10 = JSON.parse(fs.readFileSync(process.argv[10]));
    const d = await groth16.fullProve(v10, null, process.argv[8]);
    depthFromKey = d.publicSignals[2] === String(v10.nullifier) && await groth16.verify(vk10, d.publicSignals, d.proof);
  }
  console.log(JSON.stringify({ wasmFallback, depthFromKey, ms, msWarm: Math.round(msWarm * 100) / 100, publicSignals, verified, twoStepEqual, concurrentOk, batchOk, badInputRejected, unknownWasmRejected, burst, wasm: wasm ? "by sha256" : "native nLevels=160" }));
})().catch((e) => { console.error(String(e)); process.exit(1); });
