"""Synthetic zkCensus voters for fixture generation (SURVEY.md A.6 trick): the circuit only checks one
Merkle path per tree, so any sibling vector is valid once the root is recomputed from it.
Encodings follow internal/inputs.go:82-97 (decimal strings, siblings padded to nLevels+1)."""
import random, sys, os
sys.path.insert(0, os.path.dirname(__file__))
import circuit_model as cm
R = cm.R

def H(*xs):
    return cm.poseidon(cm.Rec(), 'h', [x % R for x in xs])

def climb(key, value, siblings):
    """arbo/circomlib SMT: leaf=H(key,value,1), node=H(l,r), path bit i = bit i of key; leaf sits at depth d =
    1 + index of last non-zero sibling (0 if none)."""
    d = 0
    for i, s in enumerate(siblings):
        if s: d = i + 1
    cur = H(key, value, 1)
    for i in range(d - 1, -1, -1):
        cur = H(siblings[i], cur) if (key >> i) & 1 else H(cur, siblings[i])
    return cur

def make_voter(rng, nLevels=160, depth_c=5, depth_s=8, zero_frac=0.3, address=None, avail=None, vote=None,
               eid=(102349190794087733531672488128345440122, 159684336652054988991215779568000532806)):
    address = rng.getrandbits(160) if address is None else address
    password = rng.getrandbits(88)
    signature = rng.getrandbits(512) % R
    avail = rng.randrange(1, 1000) if avail is None else avail
    vote = rng.randrange(0, avail + 1) if vote is None else vote
    def sibs(d):
        s = [0] * (nLevels + 1)
        for i in range(d):
            s[i] = 0 if (rng.random() < zero_frac and i != d - 1) else rng.randrange(1, R)
        return s
    cs, ss = sibs(depth_c), sibs(depth_s)
    sik = H(address, password, signature)
    vh = [rng.getrandbits(128), rng.getrandbits(128)]
    return {
        'electionId': [str(eid[0]), str(eid[1])],
        'nullifier': str(H(signature, password, eid[0], eid[1])),
        'availableWeight': str(avail), 'voteHash': [str(vh[0]), str(vh[1])],
        'sikRoot': str(climb(address, sik, ss)), 'censusRoot': str(climb(address, avail, cs)),
        'address': str(address), 'password': str(password), 'signature': str(signature), 'voteWeight': str(vote),
        'censusSiblings': [str(x) for x in cs], 'sikSiblings': [str(x) for x in ss],
    }
