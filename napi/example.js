// The reference's ts_inputs/src/example.ts:358-362 call, against this package (needs an MI355X and a test zkey path).
const { groth16 } = require("./index.js");
const inputs = require("../tests/golden/ref/inputs_example.json");
(async () => {
  const zkey = process.argv[2], vk = process.argv[3] ? require(require("path").resolve(process.argv[3])) : null;
  const t0 = Date.now();
  const { proof, publicSignals } = await groth16.fullProve(inputs, "../artifacts/zkCensus/dev/160/circuit.wasm", zkey);
  console.log(JSON.stringify({ ms: Date.now() - t0, publicSignals, verified: vk ? await groth16.verify(vk, publicSignals, proof) : null }));
})().catch((e) => { console.error(String(e)); process.exit(1); });
