import os, sys, ctypes, random
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import zkcensus_amd, oracle_lib as ol
from zkcensus_amd import setup
ctx = zkcensus_amd.Context(0)
_, zp, _ = setup.ensure_test_artifacts(10)
zk = open(zp,'rb').read(); pk = zkcensus_amd.ProvingKey(ctx, zk)
z = ol.zkey_parse(zk)
q, qinv = ol.Q, pow(1 << 256, -1, ol.Q)
def section(which):
    ptr, cnt, psz = {0:(z.pointsA, pk.n_vars, 64), 2:(z.pointsB2, pk.n_vars, 128)}[which]
    raw = ctypes.string_at(ptr, cnt*psz)
    return b''.join((int.from_bytes(raw[32*i:32*i+32],'little')*qinv % q).to_bytes(32,'little') for i in range(len(raw)//32)), cnt
psz_of = {0:64, 2:128}
def run(which, sc):
    std, cnt = section(which)
    scb = b''.join(x.to_bytes(32,'little') for x in sc)
    d = torch.from_numpy(np.frombuffer(scb, dtype=np.uint8).copy()).cuda()
    got = pk.msm_debug(which, d.data_ptr(), cnt)
    exp = (ol.msm_g2 if which == 2 else ol.msm_g1)(std, scb)
    if got != exp and sum(1 for x in sc if x) == 1:
        i = [k for k,x in enumerate(sc) if x][0]
        print('   got', got.hex()[:40], 'exp', exp.hex()[:40], 'pt', std[psz_of[which]*i:psz_of[which]*i+20].hex(), 'mont limb0 %08x' % ((int.from_bytes(std[psz_of[which]*i:psz_of[which]*i+32],'little') << 256) % q & 0xffffffff))
    return got == exp
cnt = pk.n_vars
# find an index whose B2 base is not infinity
std2, _ = section(2)
idx = [i for i in range(cnt) if any(std2[128*i:128*i+128])][:3]
print('nonzero B2 indices', idx)
for which in (0, 2):
    for name, vals in [('one', {idx[0]:1}), ('two', {idx[0]:2}), ('513', {idx[0]:513}), ('4097', {idx[0]:4097}), ('2^13', {idx[0]:1<<13}), ('two pts', {idx[0]:5, idx[1]:7}),
                       ('same digit 2 pts', {idx[0]:9, idx[1]:9}), ('big', {idx[0]: ol.R-1}), ('40 pts digit 3', {i:3 for i in range(idx[0], idx[0]+400)})]:
        sc = [0]*cnt
        for k,v in vals.items(): sc[k] = v
        print(which, name, run(which, sc))
