"""Synthetic voters for tests and bench (valid by construction: roots recomputed from random sibling paths,
SURVEY.md A.6; encodings per internal/inputs.go:82-97).  `H` is any Poseidon callable (list of ints -> int)."""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
EID = (102349190794087733531672488128345440122, 159684336652054988991215779568000532806)   # internal/inputs.go:60 electionId


def climb(H, key, value, siblings):
    d = 0
    for i, s in enumerate(siblings):
        if s:
            d = i + 1
    cur = H([key, value, 1])
    for i in range(d - 1, -1, -1):
        cur = H([siblings[i], cur]) if (key >> i) & 1 else H([cur, siblings[i]])
    return cur


def random_voter(rng, H, nLevels=160, depth_c=13, depth_s=13, zero_frac=0.2, avail=None, vote=None):
    address = rng.getrandbits(160)
    password = rng.getrandbits(88)
    signature = rng.getrandbits(512) % R
    avail = rng.randrange(1, 101) if avail is None else avail
    vote = rng.randrange(0, avail + 1) if vote is None else vote

    def sibs(d):
        s = [0] * (nLevels + 1)
        for i in range(d):
            s[i] = 0 if (rng.random() < zero_frac and i != d - 1) else rng.randrange(1, R)
        return s
    cs, ss = sibs(min(depth_c, nLevels)), sibs(min(depth_s, nLevels))
    sik = H([address, password, signature])
    return {
        'electionId': [str(EID[0]), str(EID[1])], 'nullifier': str(H([signature, password, EID[0], EID[1]])),
        'availableWeight': str(avail), 'voteHash': [str(rng.getrandbits(128)), str(rng.getrandbits(128))],
        'sikRoot': str(climb(H, address, sik, ss)), 'censusRoot': str(climb(H, address, avail, cs)),
        'address': str(address), 'password': str(password), 'signature': str(signature), 'voteWeight': str(vote),
        'censusSiblings': [str(x) for x in cs], 'sikSiblings': [str(x) for x in ss],
    }
