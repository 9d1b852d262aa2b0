// Type surface kept identical to what ts_inputs/src/example.ts uses from snarkjs (groth16.fullProve / prove / verify, wtns.calculate).
export interface Groth16Proof { pi_a: string[]; pi_b: string[][]; pi_c: string[]; protocol: "groth16"; curve: "bn128"; }
/** forceWasm: execute wasmFile in Node (the route a wasm without a native circuit takes by itself) even when its SHA-256 names a native witness generator */
export interface ProveOptions { nLevels?: number; r?: bigint | string; s?: bigint | string; forceWasm?: boolean; }
export type Artifact = string | Uint8Array | { type: "mem"; data?: Uint8Array };
export type CircuitInput = Record<string, string | string[]>;
export declare const groth16: {
  /** wasmFile names the circuit by its SHA-256 (dev/160 circuit.wasm -> native nLevels = 160); any other circom-2 witness calculator is executed in Node and proved on the GPU;
   *  null: ZkFranchiseProofCircuit(opts.nLevels), or the depth read off the key */
  fullProve(input: CircuitInput, wasmFile: Artifact | null, zkeyFile: Artifact, logger?: unknown, opts?: ProveOptions):
    Promise<{ proof: Groth16Proof; publicSignals: string[] }>;
  /** not in snarkjs: many voters in one call over opts.devices (one context, key and host thread per GPU); an Error entry = that voter failed a circuit assert */
  fullProveBatch(inputs: CircuitInput[], wasmFile: Artifact | null, zkeyFile: Artifact,
    opts?: { nLevels?: number; forceWasm?: boolean; devices?: number[]; rs?: Array<[bigint | string, bigint | string]> }): Promise<Array<{ proof: Groth16Proof; publicSignals: string[] } | Error>>;
  prove(zkeyFile: Artifact, wtnsFile: Artifact, logger?: unknown, opts?: ProveOptions): Promise<{ proof: Groth16Proof; publicSignals: string[] }>;
  verify(vk: object, publicSignals: string[], proof: Groth16Proof): Promise<boolean>;
};
export declare const wtns: {
  calculate(input: CircuitInput, wasmFile: Artifact | null, wtnsFile?: Artifact | null, opts?: ProveOptions): Promise<Buffer>;
};
export declare function flatten(input: CircuitInput, nLevels: number): Buffer;
