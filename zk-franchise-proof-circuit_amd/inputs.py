"""Circuit input encoding (host logic).  Mirrors the reference's generators:
internal/inputs.go:14-31,82-97 (JSON schema, decimal strings), internal/helpers.go:17-34 (BigToFF, BytesToArbo),
ts_inputs/src/inputs.ts:38-88, arbo_utils.ts:10-33, ff.ts:3-18."""
import hashlib

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
# census.circom:51-67 declaration order -- the flat order the C ABI expects
INPUT_KEYS = ['electionId', 'nullifier', 'availableWeight', 'voteHash', 'sikRoot', 'censusRoot', 'address', 'password',
              'signature', 'voteWeight', 'censusSiblings', 'sikSiblings']


def big_to_ff(x):
    """internal/helpers.go:17-26 BigToFF / ts_inputs/src/ff.ts:3-10."""
    return int(x) % R_MOD


def bytes_to_arbo(b):
    """internal/helpers.go:28-34 BytesToArbo / ts_inputs/src/arbo_utils.ts:22-33: sha256 -> two 16-byte halves, little-endian."""
    h = hashlib.sha256(bytes(b)).digest()
    return [str(int.from_bytes(h[:16], 'little')), str(int.from_bytes(h[16:], 'little'))]


def arbo_bigint(b):
    """ts_inputs/src/arbo_utils.ts:10-14: bytes read little-endian."""
    return int.from_bytes(bytes(b), 'little')


def hex_to_ff(h):
    """ts_inputs/src/ff.ts:12-18: hex string read big-endian, reduced mod r."""
    return int(h, 16) % R_MOD


def flatten_inputs(inp, nLevels=160):
    """12-key input object (or the JSON text of one: bytes / str) -> nInputs x 32-byte little-endian block.  [r5] ONE implementation for every host: the library's
    zkc_inputs_from_json (include/zkcensus.h; host only, no GPU needed), which reads the object the way circom_runtime's witness calculator does -- any key order, decimal /
    "0x" strings and integers, reduction mod r, "Signal <k> not found", "Too many values for input signal <k>", "Not all inputs have been set" -- and pads short sibling lists
    with zeros.  The N-API host (napi/index.js flatten) and the cgo host (prover.Prove's inputs []byte, zk_census_test.go:85-89) go through the same function.
    ValueError carries circom_runtime's message."""
    import ctypes, json
    from . import _native
    lib = _native.load()
    if isinstance(inp, (bytes, bytearray)):
        text = bytes(inp)
    elif isinstance(inp, str):
        text = inp.encode()
    else:
        text = json.dumps({k: v for k, v in inp.items()}, default=lambda o: str(int(o))).encode()
    out = ctypes.create_string_buffer(32 * (12 + 2 * (nLevels + 1))); err = ctypes.create_string_buffer(256)
    rc = lib.zkc_inputs_from_json(text, len(text), nLevels, out, err, 256)
    if rc:
        raise ValueError(err.value.decode())
    return out.raw
