// The reference's ts_inputs/src/example.ts:358-362 call, against this package (needs an MI355X and a test zkey):
//     node napi/example.js <zkey> [verification_key.json] [circuit.wasm]
// With a wasm path the circuit is selected by its sha256 exactly as a snarkjs caller names it; without one the native nLevels = 160
// circuit is used.  Also drives the two-step path (wtns.calculate -> groth16.prove) and four concurrent fullProve calls.
const { groth16, wtns } = require("./index.js");
const fs = require("fs");
const inputs = require("../tests/golden/ref/inputs_example.json");
(async () => {
  const zkey = process.argv[2], vk = process.argv[3] ? JSON.parse(fs.readFileSync(process.argv[3])) : null;
  const wasm = process.argv[4] && fs.existsSync(process.argv[4]) ? process.argv[4] : null;
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
  // libuv runs these on several pool threads at once: the addon serialises them on its context
  const many = await Promise.all([0, 1, 2, 3].map(() => groth16.fullProve(inputs, wasm, zkey)));
  let concurrentOk = true;
  for (const m of many) concurrentOk = concurrentOk && JSON.stringify(m.publicSignals) === JSON.stringify(publicSignals) && (!vk || await groth16.verify(vk, m.publicSignals, m.proof));
  // a batch over a pool of devices (device 0 listed twice: two contexts, two host threads): voter 1 fails an assert, the others equal fullProve with the same (r, s)
  const batch = await groth16.fullProveBatch([inputs, Object.assign({}, inputs, { nullifier: "1" }), inputs], wasm, zkey,
    { devices: [0, 0], rs: [[12345n, 67890n], [1n, 2n], [12345n, 67890n]] });
  const batchOk = batch.length === 3 && JSON.stringify(batch[0]) === JSON.stringify(b) && JSON.stringify(batch[2]) === JSON.stringify(b) &&
    batch[1] instanceof Error && /Assert Failed/.test(String(batch[1]));
  let badInputRejected = false;
  try { await groth16.fullProve(Object.assign({}, inputs, { nullifier: "1" }), wasm, zkey); } catch (e) { badInputRejected = /Assert Failed/.test(String(e)); }
  let unknownWasmRejected = false;
  try { await groth16.fullProve(inputs, Buffer.from("not a circuit"), zkey); } catch (e) { unknownWasmRejected = /unknown circuit wasm/.test(String(e)); }
  console.log(JSON.stringify({ ms, msWarm: Math.round(msWarm * 100) / 100, publicSignals, verified, twoStepEqual, concurrentOk, batchOk, badInputRejected, unknownWasmRejected, wasm: wasm ? "by sha256" : "native nLevels=160" }));
})().catch((e) => { console.error(String(e)); process.exit(1); });
