// Witness calculation by EXECUTING the caller's circom-2 witness-calculator wasm in Node's own WebAssembly -- the fallback of groth16.fullProve / wtns.calculate for a
// circuit this build has no native (HIP) witness generator for (SURVEY.md 8b: "wasmFile is accepted and hashed only to select the matching native circuit ..., falling back
// to CPU wasm execution otherwise").  The reference's caller hands snarkjs a wasm path (ts_inputs/src/example.ts:358-362) and its compiler script is written for several
// depths (circuit/circuit-compiler.sh:174-175); snarkjs runs that wasm through circom_runtime's WitnessCalculator.  This file speaks the same circom-2 ABI (SURVEY.md
// Appendix A.1/A.2), so any circom >= 2.0 circuit works: the wasm is the caller's artifact, run by the caller's runtime -- nothing of the reference ships here.
// The proof itself is still made on the GPU (groth16.prove on the resulting .wtns image, unfolded path).
"use strict";

function fnv1a64(s) {
  let h = 0xCBF29CE484222325n;
  for (let i = 0; i < s.length; i++) { h ^= BigInt(s.charCodeAt(i)); h = (h * 0x100000001B3n) & 0xFFFFFFFFFFFFFFFFn; }
  return h;
}
function flat(v, out) { if (Array.isArray(v)) for (const x of v) flat(x, out); else out.push(BigInt(v)); return out; }
// the texts circom_runtime's WitnessCalculator puts in front of the messages the wasm prints (what a snarkjs caller sees in Error.message)
const CODE_TEXT = { 1: "Signal not found.\n", 2: "Too many signals set.\n", 3: "Signal already set.\n", 4: "Assert Failed.\n", 5: "Not enough memory.\n",
  6: "Input signal array access exceeds the size.\n" };

class WasmWitnessCalculator {
  constructor(module) { this.module = module; this.inst = null; this.errStr = ""; }
  static async compile(code) { return new WasmWitnessCalculator(await WebAssembly.compile(code)); }
  async instantiate() {
    const self = this;
    const getMessage = () => { let m = "", c; while ((c = self.inst.exports.getMessageChar()) !== 0) m += String.fromCharCode(c); return m; };
    this.errStr = "";
    this.inst = await WebAssembly.instantiate(this.module, { runtime: {
      exceptionHandler(code) { const e = new Error((CODE_TEXT[code] || "Unknown error.\n") + self.errStr); e.wasmCode = code; throw e; },
      printErrorMessage() { self.errStr += getMessage() + "\n"; },
      writeBufferMessage() { getMessage(); },
      showSharedRWMemory() {},
    } });
    const ex = this.inst.exports;
    if (typeof ex.getFieldNumLen32 !== "function" || typeof ex.setInputSignal !== "function" || typeof ex.getWitness !== "function") {
      throw new Error("not a circom 2 witness calculator (exports getFieldNumLen32 / setInputSignal / getWitness are missing)");
    }
    this.n32 = ex.getFieldNumLen32();
    ex.getRawPrime();
    this.prime = 0n;
    for (let j = 0; j < this.n32; j++) this.prime |= BigInt(ex.readSharedRWMemory(j) >>> 0) << BigInt(32 * j);
    this.witnessSize = ex.getWitnessSize();
  }
  // input object -> Buffer of witnessSize x (4 n32) bytes, little-endian standard form.  Throws the Error a snarkjs caller would get (message = code text + the wasm's own lines).
  async calculate(input) {
    if (!this.inst) await this.instantiate();
    const ex = this.inst.exports, n32 = this.n32, prime = this.prime;
    this.errStr = "";
    try {
      ex.init(1);
      let set = 0;
      for (const k of Object.keys(input)) {
        const h = fnv1a64(k), hMSB = Number(h >> 32n), hLSB = Number(h & 0xFFFFFFFFn);
        const vals = flat(input[k], []);
        if (typeof ex.getInputSignalSize === "function") {
          const size = ex.getInputSignalSize(hMSB, hLSB);
          if (size < 0) throw new Error(`Signal ${k} not found\n`);
          if (size > 0 && vals.length < size) throw new Error(`Not enough values for input signal ${k}\n`);
          if (size > 0 && vals.length > size) throw new Error(`Too many values for input signal ${k}\n`);
        }
        for (let i = 0; i < vals.length; i++) {
          let v = vals[i] % prime; if (v < 0n) v += prime;
          for (let j = 0; j < n32; j++) ex.writeSharedRWMemory(j, Number((v >> BigInt(32 * j)) & 0xFFFFFFFFn));
          ex.setInputSignal(hMSB, hLSB, i);
          set++;
        }
      }
      if (typeof ex.getInputSize === "function" && set < ex.getInputSize()) throw new Error(`Not all inputs have been set. Only ${set} out of ${ex.getInputSize()}`);
      const nW = this.witnessSize, out = Buffer.alloc(nW * 4 * n32);
      for (let i = 0; i < nW; i++) {
        ex.getWitness(i);
        for (let j = 0; j < n32; j++) out.writeUInt32LE(ex.readSharedRWMemory(j) >>> 0, 4 * (i * n32 + j));
      }
      return out;
    } catch (e) {
      this.inst = null;                       // an exception leaves the instance half-run: the next call starts from a fresh one
      throw e;
    }
  }
}
// witness words (nWitness x 32 bytes LE) -> .wtns image (SURVEY.md B.1): "wtns", version 2, two sections
function wtnsImage(words, prime) {
  const n = words.length / 32, out = Buffer.alloc(12 + 12 + 40 + 12 + words.length);
  let o = 0;
  out.write("wtns", o, "latin1"); o += 4; out.writeUInt32LE(2, o); o += 4; out.writeUInt32LE(2, o); o += 4;
  out.writeUInt32LE(1, o); o += 4; out.writeBigUInt64LE(40n, o); o += 8;
  out.writeUInt32LE(32, o); o += 4;
  for (let j = 0; j < 8; j++) out.writeUInt32LE(Number((prime >> BigInt(32 * j)) & 0xFFFFFFFFn), o + 4 * j);
  o += 32; out.writeUInt32LE(n, o); o += 4;
  out.writeUInt32LE(2, o); o += 4; out.writeBigUInt64LE(BigInt(words.length), o); o += 8;
  words.copy(out, o);
  return out;
}
module.exports = { WasmWitnessCalculator, wtnsImage, fnv1a64 };
