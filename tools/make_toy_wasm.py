"""Assembles tests/golden/toy_passthrough.wasm: a hand-written circom-2-ABI witness calculator for a three-signal toy circuit,

    template Toy() { signal input a; signal input b; signal output out; out <== a; assert(a != 0); }          witness = [1, out, a, b]

It exists so that the Node surface's fallback -- groth16.fullProve / wtns.calculate EXECUTING a wasm this build has no native witness generator for (napi/wasm_witness.js) --
can be driven on the GPU box, where the reference's own circuit.wasm cannot travel.  The module is this repository's own bytes (no compiler involved: the sections are
assembled below), speaks the ABI circom 2 emits -- imports runtime.{exceptionHandler, printErrorMessage, writeBufferMessage, showSharedRWMemory}; exports getVersion ..
getMessageChar -- and raises the circom-style assert (code 4, "Error in template Toy_0 line: 7") when a == 0.

    python tools/make_toy_wasm.py            (re)writes tests/golden/toy_passthrough.wasm"""
import os
import struct

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHARED, PRIME, WIT, COUNTER, CURSOR, MSG = 0, 32, 64, 192, 200, 256          # memory map (bytes)
MESSAGE = b'Error in template Toy_0 line: 7\0'


def uleb(x):
    out = bytearray()
    while True:
        b = x & 0x7f; x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b); return bytes(out)


def sleb(x):
    out = bytearray()
    while True:
        b = x & 0x7f; x >>= 7
        if (x == 0 and not b & 0x40) or (x == -1 and b & 0x40):
            out.append(b); return bytes(out)
        out.append(b | 0x80)


def fnv1a64(s):
    h = 0xCBF29CE484222325
    for c in s.encode():
        h = ((h ^ c) * 0x100000001B3) & (2**64 - 1)
    return h


def s32(x):
    return x - (1 << 32) if x >= 1 << 31 else x


i32c = lambda v: b'\x41' + sleb(s32(v))
i64c = lambda v: b'\x42' + sleb(v)
lget = lambda i: b'\x20' + uleb(i)
ltee = lambda i: b'\x22' + uleb(i)
call = lambda f: b'\x10' + uleb(f)
i32_load = lambda off=0: b'\x28\x02' + uleb(off)
i64_load = lambda off=0: b'\x29\x03' + uleb(off)
i32_load8u = lambda off=0: b'\x2d\x00' + uleb(off)
i32_store = lambda off=0: b'\x36\x02' + uleb(off)
i64_store = lambda off=0: b'\x37\x03' + uleb(off)
IF, ELSE, END = b'\x04\x40', b'\x05', b'\x0b'
I32_EQ, I64_EQZ, I32_ADD, I32_AND, I32_OR, I32_SHL, I64_OR = b'\x46', b'\x50', b'\x6a', b'\x71', b'\x72', b'\x74', b'\x84'


def copy32(dst, src):           # 32 bytes between constant addresses
    return b''.join(i32c(0) + i32c(0) + i64_load(src + 8 * k) + i64_store(dst + 8 * k) for k in range(4))


def hash_is(name):              # (param 0, param 1) == (hMSB, hLSB) of the signal name
    h = fnv1a64(name)
    return lget(0) + i32c(h >> 32) + I32_EQ + lget(1) + i32c(h & 0xffffffff) + I32_EQ + I32_AND


def func(body, locals_i32=0):
    loc = uleb(1) + uleb(locals_i32) + b'\x7f' if locals_i32 else uleb(0)
    code = loc + body + END
    return uleb(len(code)) + code


def section(sid, payload):
    return bytes([sid]) + uleb(len(payload)) + payload


def vec(items):
    return uleb(len(items)) + b''.join(items)


def name(s):
    return uleb(len(s)) + s.encode()


def build():
    I32 = b'\x7f'
    ftype = lambda p, r: b'\x60' + vec([I32] * p) + vec([I32] * r)
    types = [ftype(0, 0), ftype(1, 0), ftype(0, 1), ftype(1, 1), ftype(2, 0), ftype(2, 1), ftype(3, 0)]
    imports = [name('runtime') + name(n) + b'\x00' + uleb(t) for n, t in (('exceptionHandler', 1), ('printErrorMessage', 0), ('writeBufferMessage', 0), ('showSharedRWMemory', 0))]
    EXC, PRINT = 0, 1
    const = lambda v: func(i32c(v))
    defs = [                                    # (export name, type index, code)
        ('getVersion', 2, const(2)), ('getMinorVersion', 2, const(1)), ('getPatchVersion', 2, const(5)), ('getSharedRWMemoryStart', 2, const(SHARED)),
        ('getFieldNumLen32', 2, const(8)),
        ('getRawPrime', 0, func(copy32(SHARED, PRIME))),
        ('getWitnessSize', 2, const(4)), ('getInputSize', 2, const(2)),
        ('getInputSignalSize', 5, func(hash_is('a') + hash_is('b') + I32_OR)),
        ('init', 1, func(i32c(0) + i64c(1) + i64_store(WIT) + b''.join(i32c(0) + i64c(0) + i64_store(WIT + 8 * k) for k in (1, 2, 3)) +
                         i32c(COUNTER) + i32c(0) + i32_store() + i32c(CURSOR) + i32c(MSG) + i32_store())),
        ('readSharedRWMemory', 3, func(lget(0) + i32c(2) + I32_SHL + i32_load(SHARED))),
        ('writeSharedRWMemory', 4, func(lget(0) + i32c(2) + I32_SHL + lget(1) + i32_store(SHARED))),
        ('setInputSignal', 6, func(
            hash_is('a') + IF +
            copy32(WIT + 64, SHARED) + copy32(WIT + 32, SHARED) +                                  # a -> wire 2, out = a -> wire 1
            i32c(0) + i64_load(0) + i32c(0) + i64_load(8) + I64_OR + i32c(0) + i64_load(16) + I64_OR + i32c(0) + i64_load(24) + I64_OR + I64_EQZ + IF +
            i32c(CURSOR) + i32c(MSG) + i32_store() + call(PRINT) + i32c(4) + call(EXC) +          # assert(a != 0)
            END +
            ELSE +
            hash_is('b') + IF + copy32(WIT + 96, SHARED) + ELSE + i32c(1) + call(EXC) + END +       # b -> wire 3 ; unknown signal: code 1
            END +
            i32c(COUNTER) + i32c(COUNTER) + i32_load() + i32c(1) + I32_ADD + i32_store())),
        ('getWitness', 1, func(b''.join(i32c(0) + lget(0) + i32c(5) + I32_SHL + i64_load(WIT + 8 * k) + i64_store(SHARED + 8 * k) for k in range(4)))),
        ('getMessageChar', 2, func(i32c(CURSOR) + i32_load() + i32_load8u() + ltee(0) + IF + i32c(CURSOR) + i32c(CURSOR) + i32_load() + i32c(1) + I32_ADD + i32_store() + END + lget(0),
                                   locals_i32=1)),
    ]
    nimp = len(imports)
    exports = [name('memory') + b'\x02' + uleb(0)] + [name(n) + b'\x00' + uleb(nimp + i) for i, (n, _, _) in enumerate(defs)]
    data = [b'\x00' + i32c(PRIME) + END + uleb(32) + R.to_bytes(32, 'little'), b'\x00' + i32c(MSG) + END + uleb(len(MESSAGE)) + MESSAGE]
    return (b'\x00asm' + struct.pack('<I', 1) + section(1, vec(types)) + section(2, vec(imports)) + section(3, vec([uleb(t) for _, t, _ in defs])) +
            section(5, vec([b'\x00' + uleb(1)])) + section(7, vec(exports)) + section(10, vec([c for _, _, c in defs])) + section(11, vec(data)))


if __name__ == '__main__':
    out = os.path.join(ROOT, 'tests', 'golden', 'toy_passthrough.wasm')
    b = build()
    open(out, 'wb').write(b)
    print(out, len(b), 'bytes')
