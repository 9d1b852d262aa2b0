// Type surface kept identical to what ts_inputs/src/example.ts uses from snarkjs.
export interface Groth16Proof { pi_a: string[]; pi_b: string[][]; pi_c: string[]; protocol: "groth16"; curve: "bn128"; }
export interface ProveOptions { nLevels?: number; r?: bigint | string; s?: bigint | string; }
export declare const groth16: {
  fullProve(input: Record<string, string | string[]>, wasmFile: string | Uint8Array | { type: "mem"; data: Uint8Array } | null,
            zkeyFile: string | Uint8Array | { type: "mem"; data: Uint8Array }, logger?: unknown, opts?: ProveOptions):
    Promise<{ proof: Groth16Proof; publicSignals: string[] }>;
  verify(vk: object, publicSignals: string[], proof: Groth16Proof): Promise<boolean>;
};
export declare function flatten(input: Record<string, string | string[]>, nLevels: number): Buffer;
