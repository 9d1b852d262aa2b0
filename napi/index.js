// zkcensus (N-API): drop-in for the snarkjs calls the reference makes (ts_inputs/src/example.ts:1,358-362):
//     const { groth16 } = require("zkcensus");   await groth16.fullProve(inputs, wasmFile, zkeyFile)
// plus groth16.prove(zkeyFile, wtnsFile), groth16.verify(vk, publicSignals, proof) and wtns.calculate(input, wasmFile, wtnsFile).
// Host code stays JavaScript/TypeScript; the arithmetic runs in libzkcensus.so's HIP kernels.  CommonJS, Node >= 12.
//
// wasmFile names the circuit, as it does for snarkjs: its SHA-256 selects the native (HIP) witness generator (80a73567...c139 = the reference's
// dev/160 circuit.wasm, artifacts/zkCensus/dev/circuits-info.md:7).  [r4] A wasm this build has no native circuit for is EXECUTED, as snarkjs would execute it
// (wasm_witness.js: the caller's circom-2 witness calculator in Node's own WebAssembly; SURVEY.md 8b), and the proof is made on the GPU from the resulting witness
// through groth16.prove's unfolded path -- any circom-2 circuit, any depth the reference's compiler script builds (circuit/circuit-compiler.sh:174-175).
// opts.forceWasm takes that route for a known wasm too.  wasmFile null/undefined selects ZkFranchiseProofCircuit(n) natively: n = opts.nLevels, or read off the key's
// header (wire count), or 160 where there is no key (wtns.calculate).
"use strict";
// hardware queues of the HIP runtime (read when it initialises, i.e. before the addon's first call): see csrc/zkc_api.hip zkc_runtime_defaults
if (!process.env.GPU_MAX_HW_QUEUES) process.env.GPU_MAX_HW_QUEUES = "24";
const fs = require("fs");
const path = require("path");
const native = require("./zkcensus.node");

const R = 21888242871839275222246405745257275088548364400416034343698204186575808495617n;
// census.circom:51-67 declaration order -- the flat order of the C ABI
const INPUT_KEYS = ["electionId", "nullifier", "availableWeight", "voteHash", "sikRoot", "censusRoot", "address", "password",
  "signature", "voteWeight", "censusSiblings", "sikSiblings"];
const LIB = process.env.ZKCENSUS_LIB || path.join(__dirname, "..", "zk-franchise-proof-circuit_amd", "libzkcensus.so");

// 32-byte little-endian image of x mod r (the blinding scalars of opts.r / opts.s), through one hex conversion
function le32(x) {
  let v = BigInt(x); if (v < 0n || v >= R) v = ((v % R) + R) % R;
  return Buffer.from(v.toString(16).padStart(64, "0"), "hex").reverse();
}
// the 12-key input object -> (12 + 2 (nLevels + 1)) x 32 bytes.  [r5] ONE implementation for every host: the library's zkc_inputs_from_json (include/zkcensus.h) reads the object
// the way circom_runtime 0.1.22's witness calculator does -- any key order, decimal / "0x" strings, integer literals and BigInts, nested arrays flattened, reduction mod r,
// "Signal <k> not found", "Too many values for input signal <k>", "Not all inputs have been set. Only <a> out of <b>" -- plus zero padding of short sibling lists; the cgo host
// passes the file image of inputs_example.json to the same function (zk_census_test.go:85-89).  JSON.stringify of 334 short strings is ~30 us.
function flatten(input, nLevels) {
  if (input === null || typeof input !== "object") throw new Error("the circuit inputs must be an object");
  let text;
  try { text = JSON.stringify(input); }                          // (a replacer is a JS call per value: 34 us of a 48 us stringify; only an object that holds BigInts needs one)
  catch (e) { if (!(e instanceof TypeError)) throw e; text = JSON.stringify(input, (k, v) => (typeof v === "bigint" ? v.toString() : v)); }
  return native.flattenJson(text, nLevels, LIB);
}
// a path is read once per (path, size, mtime): snarkjs re-reads the 55 MB .zkey on every fullProve, which here would cost more than the proof
const fileCache = new Map();
function readArtifact(f) {
  if (typeof f === "string") {
    const st = fs.statSync(f), key = `${st.size}:${st.mtimeMs}`, hit = fileCache.get(f);
    if (hit && hit.key === key) return hit.data;
    const data = fs.readFileSync(f);
    if (fileCache.size >= 4) fileCache.delete(fileCache.keys().next().value);
    fileCache.set(f, { key, data });
    return data;
  }
  // in-memory artifacts are NOT copied (Buffer.from(buffer) copies: 55 MB of key per fullProve call) and the same source object yields the same Buffer object every time,
  // so that the per-image caches -- here and in the library, which keys its SHA-256 check on (pointer, length) -- hit
  const src = f && f.type === "mem" ? f.data : f;
  if (Buffer.isBuffer(src)) return src;
  let view = memViews.get(src);
  if (!view) { view = ArrayBuffer.isView(src) ? Buffer.from(src.buffer, src.byteOffset, src.byteLength) : Buffer.from(src); if (src && typeof src === "object") memViews.set(src, view); }
  return view;
}
const memViews = new WeakMap();
const { WasmWitnessCalculator, wtnsImage } = require("./wasm_witness.js");
const wasmInfo = new WeakMap();                                 // wasm image (Buffer object) -> {nLevels, sha256}
const wasmCache = new Map();                                  // sha256 of a wasm this build executes -> its compiled calculator (at most four)
// which witness generator a call takes: {nLevels, calc}.  calc === null: the native HIP generator of ZkFranchiseProofCircuit(nLevels); else the caller's wasm, executed here
async function circuitOf(wasmFile, zkeyFile, opts) {
  const want = opts && opts.nLevels;
  if (wasmFile === null || wasmFile === undefined) {
    if (want) return { nLevels: want, calc: null };
    if (zkeyFile === null || zkeyFile === undefined) return { nLevels: 160, calc: null };
    const k = native.zkeyInfo(readArtifact(zkeyFile), LIB);
    if (k.nLevels < 0) {
      throw new Error(`no wasmFile given and the key (${k.nVars} wires, ${k.nPublic} public signals) is not a ZkFranchiseProofCircuit key: ` +
        "pass the circuit's witness-calculator wasm, or compute the witness elsewhere and call groth16.prove(zkeyFile, wtnsFile)");
    }
    return { nLevels: k.nLevels, calc: null };
  }
  const code = readArtifact(wasmFile);
  // the circuit a wasm image stands for, remembered per image OBJECT: a path yields the same cached Buffer while the file is unchanged (readArtifact), so Promise.all over a
  // census hashes the 3 MB witness calculator once, not once per ballot (2 ms each on the main thread: 500 proofs/s at best)
  let c = wasmInfo.get(code);
  if (!c) { c = native.circuitFromWasm(code, LIB); wasmInfo.set(code, c); }
  if (c.nLevels >= 0 && !(opts && opts.forceWasm)) {
    if (want && want !== c.nLevels) throw new Error(`wasm file is the nLevels=${c.nLevels} circuit but nLevels=${want} was requested`);
    return { nLevels: c.nLevels, calc: null };
  }
  let calc = wasmCache.get(c.sha256);
  if (!calc) {
    try { calc = await WasmWitnessCalculator.compile(code); await calc.instantiate(); }
    catch (e) { throw new Error(`unknown circuit wasm (sha256 ${c.sha256}) and it cannot be executed as a circom 2 witness calculator: ${e.message}`); }
    if (wasmCache.size >= 4) wasmCache.delete(wasmCache.keys().next().value);
    wasmCache.set(c.sha256, calc);
  }
  return { nLevels: c.nLevels, calc };
}
function toJson(out) {
  const d = native.decimals(out.proof);                       // 8 x 32-byte words -> decimal strings (C++: Node's BigInt -> decimal was 4 us a value)
  const proof = { pi_a: [d[0], d[1], "1"], pi_b: [[d[2], d[3]], [d[4], d[5]], ["1", "0"]], pi_c: [d[6], d[7], "1"],
    protocol: "groth16", curve: "bn128" };
  return { proof, publicSignals: native.decimals(out.publicSignals) };
}
const blind = (opts, k) => (opts && opts[k] !== undefined ? le32(opts[k]) : null);      // null: drawn uniformly in Fr by the library

const wtns = {
  // snarkjs wtns.calculate(input, wasmFile, wtnsFileName): writes the .wtns file (or fills {type: "mem"}.data); also returns the image
  async calculate(input, wasmFile, wtnsFile, opts) {
    const c = await circuitOf(wasmFile, null, opts);
    const image = c.calc ? wtnsImage(await c.calc.calculate(input), c.calc.prime) : await native.witnessRaw(flatten(input, c.nLevels), c.nLevels, LIB);
    if (typeof wtnsFile === "string") fs.writeFileSync(wtnsFile, image);
    else if (wtnsFile && wtnsFile.type === "mem") wtnsFile.data = new Uint8Array(image);
    return image;
  },
};
const groth16 = {
  async fullProve(input, wasmFile, zkeyFile, logger, opts) {
    const c = await circuitOf(wasmFile, zkeyFile, opts);
    if (c.calc) return toJson(await native.proveRaw(readArtifact(zkeyFile), wtnsImage(await c.calc.calculate(input), c.calc.prime), blind(opts, "r"), blind(opts, "s"), LIB));
    return toJson(await native.fullProveRaw(flatten(input, c.nLevels), c.nLevels, readArtifact(zkeyFile), blind(opts, "r"), blind(opts, "s"), LIB));
  },
  // Not in snarkjs: a census worth of voters in one call, split over opts.devices (default [0]) -- one context, resident key and host thread per GPU
  // (zkc_pool_* in include/zkcensus.h).  Resolves to one entry per voter, in order: {proof, publicSignals}, or an Error for a voter whose inputs fail
  // a circuit assert (the others are unaffected).  opts.rs: [[r, s], ...] per voter for reproducible bytes.
  async fullProveBatch(inputs, wasmFile, zkeyFile, opts) {
    const c = await circuitOf(wasmFile, zkeyFile, opts);
    if (c.calc) {                 // a circuit without a native witness generator: witnesses one by one in Node, proofs coalesced by the library's proving service
      // the key image is read ONCE (a Buffer / {type: "mem"} artifact is copied by readArtifact: one copy per voter was 56 GB of host memory for a census of 1 024 and a new
      // (pointer, length) -- a new SHA-256 over the whole key -- per request), and the voters go through in chunks: witnesses of a chunk are computed (main thread, Node's
      // WebAssembly) and submitted, the next chunk's witnesses run while the GPU proves
      const zk = readArtifact(zkeyFile), out = new Array(inputs.length), CHUNK = 64;
      let pending = [];
      for (let i0 = 0; i0 < inputs.length; i0 += CHUNK) {
        const chunk = [];
        for (let i = i0; i < Math.min(inputs.length, i0 + CHUNK); i++) {
          chunk.push((async () => {
            try {
              const o = opts && opts.rs ? { r: opts.rs[i][0], s: opts.rs[i][1] } : null;
              out[i] = toJson(await native.proveRaw(zk, wtnsImage(await c.calc.calculate(inputs[i]), c.calc.prime), blind(o, "r"), blind(o, "s"), LIB));
            } catch (e) { out[i] = e instanceof Error ? e : new Error(String(e)); }
          })());
        }
        await Promise.all(pending);                              // at most two chunks of witnesses and proofs alive at once
        pending = chunk;
      }
      await Promise.all(pending);
      return out;
    }
    const nLevels = c.nLevels;
    const devs = Buffer.alloc(4 * ((opts && opts.devices) || [0]).length);
    ((opts && opts.devices) || [0]).forEach((d, i) => devs.writeInt32LE(d, 4 * i));
    const rs = opts && opts.rs ? Buffer.concat(opts.rs.map(([r, s]) => Buffer.concat([le32(r), le32(s)]))) : null;
    const out = await native.fullProveBatchRaw(Buffer.concat(inputs.map((x) => flatten(x, nLevels))), nLevels, readArtifact(zkeyFile), devs, rs, LIB);
    const np = out.publicSignals.length / inputs.length;
    return inputs.map((_, i) => {
      const st = out.status.readInt32LE(4 * i);
      if (st !== 0) return new Error(native.statusText(nLevels, st, LIB));
      return toJson({ proof: out.proofs.subarray(256 * i, 256 * i + 256), publicSignals: out.publicSignals.subarray(np * i, np * i + np) });
    });
  },
  async prove(zkeyFile, wtnsFile, logger, opts) {
    return toJson(await native.proveRaw(readArtifact(zkeyFile), readArtifact(wtnsFile), blind(opts, "r"), blind(opts, "s"), LIB));
  },
  async verify(vk, publicSignals, proof) {
    return native.verifyJson(JSON.stringify(vk), JSON.stringify(publicSignals), JSON.stringify(proof), LIB);
  },
};
module.exports = { groth16, wtns, flatten };
