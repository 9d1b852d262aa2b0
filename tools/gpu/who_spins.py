#!/usr/bin/env python3
"""Which host threads burn CPU while the batch pipeline runs?  Lists the threads each start-up stage creates, then samples every thread's state / syscall /
CPU time while batch steps run, and finally lets rocgdb print the top frames of the busy ones."""
import os, sys, time, threading, subprocess, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')

def tids():
    return set(os.listdir('/proc/self/task'))

def cpu(tid):
    try:
        f = open('/proc/self/task/%s/stat' % tid).read(); v = f[f.rindex(')') + 2:].split(); return v[0], (int(v[11]) + int(v[12])) / os.sysconf('SC_CLK_TCK')
    except OSError:
        return '?', 0.0

seen = tids(); print('start', len(seen))
def stage(name):
    global seen
    now = tids(); print('%-28s new threads: %s' % (name, sorted(now - seen, key=int))); seen = now
import numpy as np; stage('numpy')
import torch; stage('import torch')
torch.cuda.init(); x = torch.zeros(4, device='cuda'); stage('torch cuda init')
import zkcensus_amd
from zkcensus_amd import setup, census
ctx = zkcensus_amd.Context(0); stage('zkc context')
_, zkey_path, _ = setup.ensure_test_artifacts(160)
pk = zkcensus_amd.ProvingKey(ctx, open(zkey_path, 'rb').read()); stage('key load')
B = 1024
flat, _, _ = census.synthetic_census_flat(ctx, 8192, 160); stage('census')
nIn = ctx.n_inputs(160); nW = ctx.n_wires(160)
d_in = torch.from_numpy(np.frombuffer(flat[:B * nIn * 32], dtype=np.uint8).copy()).cuda()
d_w = [torch.empty(B * nW * 32, dtype=torch.uint8, device='cuda') for _ in range(2)]; d_s = [torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(2)]
rs = os.urandom(0) + b''.join((i * 7919 + 11).to_bytes(32, 'little') for i in range(2 * B))
pk.batch_begin(0, d_in.data_ptr(), B, d_w[0].data_ptr(), d_s[0].data_ptr(), rs); pk.batch_finish(0, B); stage('first step')
stop = False
def run():
    i = 0
    pk.batch_begin(0, d_in.data_ptr(), B, d_w[0].data_ptr(), d_s[0].data_ptr(), rs)
    while not stop:
        i ^= 1
        pk.batch_begin(i, d_in.data_ptr(), B, d_w[i].data_ptr(), d_s[i].data_ptr(), rs)
        pk.batch_finish(i ^ 1, B)
    pk.batch_finish(i, B)
th = threading.Thread(target=run); th.start()
time.sleep(1.0); stage('steps running')
c0 = {t: cpu(t)[1] for t in tids()}; w0 = time.time(); samples = {}
for _ in range(40):
    time.sleep(0.05)
    for t in tids():
        st, _c = cpu(t)
        try: sc = open('/proc/self/task/%s/syscall' % t).read().split()[0]
        except OSError: sc = '?'
        samples.setdefault(t, []).append(st + ':' + sc)
dt = time.time() - w0
busy = sorted(((cpu(t)[1] - c0.get(t, 0)) / dt, t) for t in tids())[-4:]
for u, t in reversed(busy):
    s = samples.get(t, []); print('tid %s  %.2f cores  states %s' % (t, u, {k: s.count(k) for k in set(s)}))
if os.environ.get('WHO_GDB') == '1':
    libc = ctypes.CDLL(None); libc.prctl(0x59616d61, ctypes.c_ulong(-1 & 0xffffffffffffffff), 0, 0, 0)      # PR_SET_PTRACER, PR_SET_PTRACER_ANY
    cmds = []
    for u, t in reversed(busy[-3:]):
        cmds += ['-ex', 'thread find %s' % t]
    r = subprocess.run(['/opt/rocm/bin/rocgdb', '-p', str(os.getpid()), '-batch', '-ex', 'set pagination off', '-ex', 'thread apply all bt 6'], capture_output=True, text=True, timeout=240)
    out = r.stdout
    for u, t in reversed(busy[-3:]):
        k = out.find('LWP %s)' % t)
        print('---- tid %s (%.2f cores)\n%s' % (t, u, out[max(0, k - 60):k + 900] if k >= 0 else '(not found)'))
    if r.returncode: print(r.stderr[-600:])
stop = True; th.join(); pk.close(); ctx.close()
