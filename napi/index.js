// zkcensus (N-API): drop-in for the two snarkjs calls the reference makes (ts_inputs/src/example.ts:1,358-362):
//     const { groth16 } = require("zkcensus");   await groth16.fullProve(inputs, wasmFile, zkeyFile)
// Host code stays JavaScript/TypeScript; the arithmetic runs in libzkcensus.so's HIP kernels.  CommonJS, Node >= 12.
"use strict";
const fs = require("fs");
const path = require("path");
const crypto = require("crypto");
const native = require("./zkcensus.node");

const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
// census.circom:51-67 declaration order -- the flat order of the C ABI
const INPUT_KEYS = ["electionId", "nullifier", "availableWeight", "voteHash", "sikRoot", "censusRoot", "address", "password",
  "signature", "voteWeight", "censusSiblings", "sikSiblings"];
const LIB = process.env.ZKCENSUS_LIB || path.join(__dirname, "..", "zk-franchise-proof-circuit_amd", "libzkcensus.so");

function le32(x) { const b = Buffer.alloc(32); let v = ((BigInt(x) % R) + R) % R; for (let i = 0; i < 32; i++) { b[i] = Number(v & 0xffn); v >>= 8n; } return b; }
function fromLe(b, off) { let v = 0n; for (let i = 31; i >= 0; i--) v = (v << 8n) | BigInt(b[off + i]); return v; }
function flatten(input, nLevels) {
  const parts = [];
  for (const k of INPUT_KEYS) {
    if (!(k in input)) throw new Error(`Error: Signal not found.\n(input ${k})`);
    let v = input[k];
    if (k.endsWith("Siblings")) {
      v = Array.from(v);
      if (v.length > nLevels + 1) throw new Error(`Too many values for input signal ${k}`);
      while (v.length < nLevels + 1) v.push("0");
    }
    for (const x of (Array.isArray(v) ? v : [v])) parts.push(le32(x));
  }
  return Buffer.concat(parts);
}
function readArtifact(f) {
  if (typeof f === "string") return fs.readFileSync(f);
  if (f && f.type === "mem") return Buffer.from(f.data);
  return Buffer.from(f);
}
function rand32() { const b = crypto.randomBytes(32); b[31] = 0; return b; }     // < 2^248 < field order

const groth16 = {
  // wasmFile is accepted for source compatibility; the witness is computed natively for the zkCensus circuit
  async fullProve(input, wasmFile, zkeyFile, logger, opts) {
    const nLevels = (opts && opts.nLevels) || 160;
    const r = opts && opts.r !== undefined ? le32(opts.r) : rand32(), s = opts && opts.s !== undefined ? le32(opts.s) : rand32();
    const out = await native.fullProveRaw(flatten(input, nLevels), nLevels, readArtifact(zkeyFile), r, s, LIB);
    const p = out.proof, d = (o) => fromLe(p, o).toString();
    const proof = { pi_a: [d(0), d(32), "1"], pi_b: [[d(64), d(96)], [d(128), d(160)], ["1", "0"]], pi_c: [d(192), d(224), "1"],
      protocol: "groth16", curve: "bn128" };
    const publicSignals = [];
    for (let i = 0; i < out.publicSignals.length / 32; i++) publicSignals.push(fromLe(out.publicSignals, 32 * i).toString());
    return { proof, publicSignals };
  },
  async verify(vk, publicSignals, proof) {
    return native.verifyJson(JSON.stringify(vk), JSON.stringify(publicSignals), JSON.stringify(proof), LIB);
  },
};
module.exports = { groth16, flatten };
