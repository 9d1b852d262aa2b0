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
    """12-key input object -> nInputs x 32-byte little-endian block (values reduced mod r, as circom_runtime does)."""
    out = []
    for k in INPUT_KEYS:
        v = inp[k]
        if k.endswith('Siblings'):
            v = list(v)
            if len(v) > nLevels + 1:
                raise ValueError('%s: too many values for input signal (%d > %d)' % (k, len(v), nLevels + 1))
            v = v + ['0'] * (nLevels + 1 - len(v))
        if isinstance(v, (list, tuple)):
            out.extend(int(x) % R_MOD for x in v)
        else:
            out.append(int(v) % R_MOD)
    if len(out) != 12 + 2 * (nLevels + 1):
        raise ValueError('Not all inputs have been set: %d of %d' % (len(out), 12 + 2 * (nLevels + 1)))
    return b''.join(x.to_bytes(32, 'little') for x in out)
