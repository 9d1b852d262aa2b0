#!/usr/bin/env python3
"""Recover circom 2.1.5's wire order for the zkCensus circuit by value-matching the wasm oracle's witnesses
(tools/wasm_witness.js) against tools/circuit_model.py on several inputs, under the hypothesis
  wires = [one, public main signals, private main signals] ++ DFS(components sorted by name,
          each component's own surviving signals first: outputs, inputs, intermediates in declaration order).
Greedy in-order matching succeeds only if the hypothesis explains every wire.  Prints the survivor list in a
run-length-compressed generic form.  Exploration tool (build container only)."""
import sys, json, collections, re, os, pickle
sys.path.insert(0, os.path.dirname(__file__))
import circuit_model as cm

SIG = {
 'checkWeight': ['out', 'in'], 'lt': ['out', 'in'], 'n2b': ['out', 'in'], 'num2bits': ['out', 'in'],
 'sik': ['out', 'inputs'], 'computedNullifier': ['out', 'inputs'], 'h': ['out', 'inputs'],
 'pEx': ['out', 'inputs', 'initialState'], 'ark': ['out', 'in'], 'mix': ['out', 'in'], 'mixS': ['out', 'in'],
 'mixLast': ['out', 'in'], 'sigmaF': ['out', 'in', 'in2', 'in4'], 'sigmaP': ['out', 'in', 'in2', 'in4'],
 'sikVerifier': ['enabled', 'root', 'siblings', 'oldKey', 'oldValue', 'isOld0', 'key', 'value', 'fnc'],
 'censusVerifier': ['enabled', 'root', 'siblings', 'oldKey', 'oldValue', 'isOld0', 'key', 'value', 'fnc'],
 'hash1New': ['out', 'key', 'value'], 'hash1Old': ['out', 'key', 'value'], 'proofHash': ['out', 'L', 'R'],
 'n2bNew': ['out', 'in'], 'n2bOld': ['out', 'in'], 'aliasCheck': ['in'], 'compConstant': ['out', 'in', 'parts', 'sout'],
 'smtLevIns': ['levIns', 'enabled', 'siblings', 'done'], 'isZero': ['out', 'in', 'inv'], 'isz': ['out', 'in', 'inv'],
 'sm': ['st_top', 'st_i0', 'st_iold', 'st_inew', 'st_na', 'is0', 'levIns', 'fnc', 'prev_top', 'prev_i0', 'prev_iold', 'prev_inew',
        'prev_na', 'prev_top_lev_ins', 'prev_top_lev_ins_fnc'],
 'levels': ['root', 'st_top', 'st_i0', 'st_iold', 'st_inew', 'st_na', 'sibling', 'old1leaf', 'new1leaf', 'lrbit', 'child', 'aux'],
 'switcher': ['outL', 'outR', 'sel', 'L', 'R', 'aux'], 'areKeyEquals': ['out', 'in'], 'checkRoot': ['enabled', 'in'],
 'checkNullifier': ['enabled', 'in'],
}
tok = re.compile(r'([A-Za-z_0-9]+)((?:\[\d+\])*)')
def parse(part):
    m = tok.fullmatch(part); return m.group(1), tuple(int(x) for x in re.findall(r'\[(\d+)\]', m.group(2)))
def sort_key(name, main_pos):
    parts = name.split('.')
    if parts[0] == 'one': return ((0, 0),)
    if len(parts) == 2: return ((0, 1, main_pos[name]),)
    key = []
    for comp in parts[1:-1]:
        b, idx = parse(comp); key.append((1, b, idx))
    b, idx = parse(parts[-1])
    kind = parse(parts[-2])[0]
    key.append((0, SIG[kind].index(b), idx))
    return tuple(key)

def main():
    inputs = [json.load(open('/root/reference/artifacts/zkCensus/dev/160/inputs_example.json'))] + json.load(open('/tmp/w/voters.json'))
    files = ['/tmp/w/example.bin'] + ['/tmp/w/v.bin.%d' % i for i in range(len(inputs) - 1)]
    cand = None
    for inp, f in zip(inputs, files):
        rec = cm.census_circuit(inp); names = rec.names
        w = open(f, 'rb').read(); W = [int.from_bytes(w[32 * i:32 * i + 32], 'little') for i in range(len(w) // 32)]
        byval = collections.defaultdict(set)
        for i, v in enumerate(rec.vals): byval[v].add(i)
        cand = [set(byval[v]) for v in W] if cand is None else [c & byval[v] for c, v in zip(cand, W)]
    pub = ['electionId[0]', 'electionId[1]', 'nullifier', 'voteHash[0]', 'voteHash[1]', 'sikRoot', 'censusRoot', 'voteWeight',
           'availableWeight', 'address', 'password', 'signature'] + ['censusSiblings[%d]' % i for i in range(161)] + ['sikSiblings[%d]' % i for i in range(161)]
    main_pos = {'main.' + n: i for i, n in enumerate(pub)}
    order = sorted(range(len(names)), key=lambda i: sort_key(names[i], main_pos))
    p = 0; chosen = []
    for wi, c in enumerate(cand):
        q = p
        while q < len(order) and order[q] not in c: q += 1
        if q == len(order):
            print('FAIL at wire', wi, 'prev', names[chosen[-1]] if chosen else None, 'cands', [names[j] for j in sorted(c)[:6]]); break
        chosen.append(order[q]); p = q + 1
    else:
        print('hypothesis explains all', len(cand), 'wires')
    pickle.dump([names[j] for j in chosen], open('/tmp/w/chosen.pkl', 'wb'))
    # compressed print
    gen = lambda n: re.sub(r'\d+', '#', n)
    prev = None; cnt = 0; start = 0
    for i, j in enumerate(chosen + [None]):
        g = gen(names[j]) if j is not None else None
        if g != prev:
            if prev is not None: print('%6d x%-4d %s   e.g. %s' % (start, cnt, prev, names[chosen[start]]))
            prev = g; cnt = 0; start = i
        cnt += 1

if __name__ == '__main__':
    main()
