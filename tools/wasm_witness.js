#!/usr/bin/env node
// Drives a circom-2.x witness-calculator wasm (the reference's committed artifact
// artifacts/zkCensus/dev/160/circuit.wasm) and dumps the witness as nWires x 32-byte LE words.
// This loader is this repo's own code (ABI described in SURVEY.md Appendix A); it is a
// fixture-generation tool for THIS container only -- nothing in tests/ or the product runs it.
//
// usage: node tools/wasm_witness.js <circuit.wasm> <inputs.json> <out.bin>   (prints JSON status line)
"use strict";
const fs = require("fs");

function fnv1a64(s) {
  let h = 0xCBF29CE484222325n;
  for (let i = 0; i < s.length; i++) {
    h ^= BigInt(s.charCodeAt(i));
    h = (h * 0x100000001B3n) & 0xFFFFFFFFFFFFFFFFn;
  }
  return h;
}
function flatten(v, out) { if (Array.isArray(v)) v.forEach(x => flatten(x, out)); else out.push(BigInt(v)); return out; }

async function load(wasmPath) {
  const code = fs.readFileSync(wasmPath);
  let inst, errStr = "";
  const getMessage = () => { let m = ""; let c; while ((c = inst.exports.getMessageChar()) !== 0) m += String.fromCharCode(c); return m; };
  const mod = await WebAssembly.instantiate(code, { runtime: {
    exceptionHandler(code) { const e = new Error("wasm exception code " + code + ": " + errStr); e.code = code; e.msg = errStr; throw e; },
    printErrorMessage() { errStr += getMessage() + "\n"; },
    writeBufferMessage() { getMessage(); },
    showSharedRWMemory() {},
  }});
  inst = mod.instance;
  return { inst, resetErr() { errStr = ""; } };
}

function run(ctx, inputs) {
  const ex = ctx.inst.exports;
  ctx.resetErr();
  const n32 = ex.getFieldNumLen32();
  ex.getRawPrime();
  let prime = 0n;
  for (let j = 0; j < n32; j++) prime |= BigInt(ex.readSharedRWMemory(j) >>> 0) << BigInt(32 * j);
  ex.init(1);
  for (const k of Object.keys(inputs)) {
    const h = fnv1a64(k);
    const hMSB = Number(h >> 32n), hLSB = Number(h & 0xFFFFFFFFn);
    const vals = flatten(inputs[k], []);
    for (let i = 0; i < vals.length; i++) {
      let v = ((vals[i] % prime) + prime) % prime;
      for (let j = 0; j < n32; j++) ex.writeSharedRWMemory(j, Number((v >> BigInt(32 * j)) & 0xFFFFFFFFn));
      ex.setInputSignal(hMSB, hLSB, i);
    }
  }
  const nW = ex.getWitnessSize();
  const out = Buffer.alloc(nW * 32);
  for (let i = 0; i < nW; i++) {
    ex.getWitness(i);
    for (let j = 0; j < n32; j++) out.writeUInt32LE(ex.readSharedRWMemory(j) >>> 0, i * 32 + 4 * j);
  }
  return out;
}

module.exports = { load, run };

if (require.main === module) (async () => {
  const [wasmPath, inPath, outPath] = process.argv.slice(2);
  const ctx = await load(wasmPath);
  // inPath may hold one input object or an array of them; outPath gets .<i> suffix for arrays
  const j = JSON.parse(fs.readFileSync(inPath, "utf8"));
  const list = Array.isArray(j) ? j : [j];
  const status = [];
  for (let i = 0; i < list.length; i++) {
    try {
      const t0 = Date.now();
      const w = run(ctx, list[i]);
      fs.writeFileSync(Array.isArray(j) ? outPath + "." + i : outPath, w);
      status.push({ ok: true, ms: Date.now() - t0, sha256: require("crypto").createHash("sha256").update(w).digest("hex") });
    } catch (e) {
      status.push({ ok: false, code: e.code, msg: (e.msg || String(e)).trim() });
    }
  }
  console.log(JSON.stringify(status));
})().catch(e => { console.error(e); process.exit(1); });
