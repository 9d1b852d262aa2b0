#!/usr/bin/env python3
"""bench.py -- zkCensus proofs/sec on MI355X (BASELINE.json metric), one process per GPU.

A step = one pass of the hot path (witness -> buildABC -> NTT -> 5 MSMs -> blinding) over one batch of --batch synthetic
voters whose 334 x 32-byte input blocks are already resident in HBM; with N > 1 every rank proves its own block of the
census (weak scaling) and the finished 512-byte proofs are gathered to every rank with RCCL inside the timed region.
`python bench.py --gpus N` launches its own N ranks when it was not started by torch.distributed.run (the parent never touches
the GPU; the ranks are child processes).  Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel's ALGORITHMIC bytes / its HIP-event duration on the library's stream vs 8 TB/s
  cpu_baseline : the CPU oracle (oracle/, the build's own port) on all host cores, one proof per thread, on a bounded sample of
                 the same workload -- and every one of those oracle proofs is byte-compared with the GPU proof of the same voter
  verified     : what was checked about the proofs of the LAST TIMED step (batch verifier over all of them, oracle verifier
                 and oracle prover on samples that include the pass boundaries).
"""
import argparse, ctypes, json, os, socket, subprocess, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# hardware queues for the HIP runtime of this process (read once, when the runtime initialises -- before torch or the library touches the GPU): the pipeline's seven streams
# on the default four queues run 5 % slower (csrc/zkc_api.hip zkc_runtime_defaults; profiles/r05_hw_queues_ab.json)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
# the host side of a step is a few kilobytes of numpy / torch bookkeeping: one thread each.  Left alone, OpenBLAS starts 64 threads at import and torch's intra-op pool one per
# schedulable core (256 on the GPU boxes) -- and eight ranks share 16 cores' worth of CPU time (profiles/r04_cpu_baseline_scaling.json); a parallel region that wakes 256 threads to add
# up 1 024 status words costs more than the step's enqueueing.  (The CPU baseline leg and the batch verifier use their own threads, not these pools.)
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, '1')
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.3 TB/s achievable)
CATS = {0: 'witness', 1: 'buildABC_matvec', 2: 'ntt_joinABC', 3: 'msm_digits_sort', 4: 'msm_accumulate_g1', 5: 'msm_accumulate_g2', 6: 'msm_reduce',
        7: 'msm_g1_streamed'}
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=int(os.environ.get('ZKC_BATCH', '1024')), help='voter proofs per GPU per step')
    ap.add_argument('--nlevels', type=int, default=160)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-verify', action='store_true', help='skip the post-run verification of the timed proofs')
    ap.add_argument('--no-step-pipelining', action='store_true', help='one synchronous call per step (rounds 1-2) instead of begin(k + 1) before finish(k)')
    ap.add_argument('--no-extras', action='store_true', help='skip the extra legs after the timed region (host-to-host rate, unfolded / worst-case rates, isolated per-stage times)')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='launcher / rendezvous / gather rehearsal on the CPU (gloo, fabricated records, no prover): NOT a measurement')
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a launcher: start N ranks of this script as CHILD processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, exactly what torch.distributed.run would set).  The parent never initialises HIP, so nothing is exec'ed over a process
    that has touched the GPU; it relays rank 0's JSON line and exits with the worst child status."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is read on a thread so that the parent can watch ALL children: if one rank dies, the others would sit in the collective until
    # RCCL's own timeout; they are terminated (by their exact pids) instead and the failure is reported at once
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True); reader.start()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() is not None and p.returncode != 0]
        if bad:
            failed = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill(); p.wait()
    reader.join(timeout=10)
    # only the JSON line goes to stdout; anything else rank 0 printed there (a library banner such as gloo's "[Gloo] Rank 0 is connected ...") to stderr
    for ln in b''.join(c for c in chunks if c).decode(errors='replace').splitlines():
        (sys.stdout if ln.lstrip().startswith('{') else sys.stderr).write(ln + '\n')
    sys.stdout.flush()
    if failed:
        sys.stderr.write('bench.py: rank %d exited with status %d; the other ranks were terminated\n' % failed)
    rcs = [p.returncode for p in procs]
    return max(abs(rc) for rc in rcs)


def draw_rs(rng, n):
    """n blinding scalars uniform in [0, r): 254 random bits, rejected when >= r (what snarkjs' Fr.random / rapidsnark do), 32 B LE each."""
    import numpy as np
    lim = np.frombuffer(R_MOD.to_bytes(32, 'little'), dtype='<u8')
    out = np.empty((n, 32), dtype=np.uint8); have = 0
    while have < n:
        a = rng.integers(0, 256, size=(2 * (n - have) + 8, 32), dtype=np.uint8); a[:, 31] &= 0x3f
        q = a.view('<u8')
        lt = q[:, 3] < lim[3]
        for k in (2, 1, 0):
            eq = np.ones(len(q), dtype=bool)
            for j in range(3, k, -1):
                eq &= q[:, j] == lim[j]
            lt |= eq & (q[:, k] < lim[k])
        good = a[lt][:n - have]
        out[have:have + len(good)] = good; have += len(good)
    return out


def dry_run(args, rank, world):
    """CPU rehearsal of the N > 1 control path: same launcher, rendezvous, shard_range / pack_records / gather_records and
    max-over-ranks timing as the real run, with fabricated 513-byte records instead of proofs.  The line it prints is marked invalid."""
    import torch, torch.distributed as dist
    from zkcensus_amd import parallel
    if os.environ.get('ZKC_BENCH_TEST_FAIL_RANK') == str(rank):          # test hook: a rank that dies must not leave the launcher hanging
        raise SystemExit(7)
    if world > 1:
        dist.init_process_group('gloo')
    B = min(args.batch, 64); total = B * world
    if os.environ.get('ZKC_BENCH_TEST_UNEVEN') == '1':                   # test hook: a census that does not divide by the ranks (the last ranks prove one voter less)
        total -= world // 2 + 1
    lo, hi = parallel.shard_range(rank, world, total)
    # the census hand-out of the real run: rank 0 holds every voter's input block, one broadcast, each rank keeps its slice
    blk = 64
    allin = torch.tensor([(v * 13 + k) % 251 for v in range(total) for k in range(blk)], dtype=torch.uint8) if rank == 0 else torch.empty(total * blk, dtype=torch.uint8)
    if world > 1:
        dist.broadcast(allin, src=0)
    census_ok = allin[lo * blk:hi * blk].tolist() == [(v * 13 + k) % 251 for v in range(lo, hi) for k in range(blk)]
    fab = lambda v, m: bytes((v * 7 + k * m) % 251 for k in range(256))
    rec = parallel.pack_records(b''.join(fab(v, 1) for v in range(lo, hi)), b''.join(fab(v, 3) for v in range(lo, hi)), [0] * (hi - lo))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allrec = parallel.gather_records(rec, world, dist, total) if world > 1 else rec
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    each = [tmax.clone()]
    if world > 1:
        each = [torch.empty_like(tmax) for _ in range(world)]
        dist.all_gather(each, tmax.clone())
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ok = census_ok and allrec.shape[0] == total and all(bytes(allrec[v, :256].tolist()) == fab(v, 1) for v in range(total)) and bool((allrec[lo:hi] == rec).all())
    okt = torch.tensor([1 if ok else 0])
    if world > 1:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({'metric': 'DRY RUN (CPU, gloo, fabricated records) -- not a measurement', 'value': None, 'unit': 'proofs/s', 'n_gpus': world,
                          'steps': args.steps, 'warmup': args.warmup, 'dry_run': True, 'valid': False, 'gathered_records_equal_per_rank_records': bool(okt.item()),
                          'voters': total, 'ms_per_step_per_rank': [round(float(t.item()) / max(1, args.steps) * 1e3, 3) for t in each]}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if okt.item() else 1


def cores_received(ol, nthreads, seconds=0.4):
    """the cores' worth of CPU time this container actually gets when it asks for `nthreads`: the C oracle's own NTT (2^13 points, a few ms per call, ctypes releases the GIL)
    on that many threads, process CPU seconds over wall seconds.  About nthreads on an unconstrained host, the CPU-time quota on a constrained one.  (cpu_baseline leg only.)"""
    import threading
    lib = ol.lib(); stop = time.perf_counter() + seconds

    def spin():
        d = (ctypes.c_uint64 * (4 << 13))()
        while time.perf_counter() < stop:
            lib.zko_ntt(d, 13, 0)
    th = [threading.Thread(target=spin) for _ in range(nthreads)]
    t0 = os.times(); w0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    t1 = os.times(); wall = time.perf_counter() - w0
    return ((t1.user - t0.user) + (t1.system - t0.system)) / wall


def read_prof(ctx):
    prof = {}
    for cat, name in CATS.items():
        ms, n, by = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        ctx._lib.zkc_profile_read(ctx._h, cat, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(by))
        prof[name] = {'ms': ms.value, 'launches': n.value, 'alg_bytes': by.value}
    return prof


def extra_legs(args, ctx, pk, zkey_bytes, vk, B, d_inputs, d_wtns, d_status, flat, out, rs, madds_folded):
    """What the headline does not say, measured after the timed region on rank 0 at N = 1 (each leg 1 warm-up + 2 timed calls, seconds in all):
      host_to_host     the same step with the input blocks in pinned HOST memory when the clock starts (H2D inside the step; SURVEY.md 8d config 3's wording)
      folding          the headline relies on constant folding (a real census puts leaves 13-17 levels deep, every level below carries the voter-independent
                       empty-subtree trace).  (a) voters whose leaves sit at the very bottom of both trees, through the same key: nothing folds; (b) the same
                       witnesses through a key loaded with ZKC_NO_FOLD=1, witness given -- the groth16.prove(zkey, wtns) path for a foreign witness -- with
                       bytes compared against the folded proofs of the timed step
      single_proof     the first voter of the batch alone: inputs -> proof latency of one call (BASELINE configs[1]), and that its bytes equal the batch's
      stages_isolated  one pass with every stage on ONE stream (ZKC_SERIAL_STREAMS=1): the HIP-event brackets per kernel category are then isolated kernel
                       times, priced against their algorithmic bytes and the 8 TB/s roof"""
    import numpy as np, torch
    import zkcensus_amd
    from zkcensus_amd import census, groth16
    res = {}
    nW = pk.n_vars; nl = args.nlevels
    # ---- host to host ----
    h_in = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).pin_memory()

    def h2h():
        d_inputs.copy_(h_in)
        return pk.fullprove_batch_dev(d_inputs.data_ptr(), B, d_wtns.data_ptr(), d_status.data_ptr(), draw_rs(rs, 2 * B).tobytes())
    h2h(); torch.cuda.synchronize(); t0 = time.perf_counter(); h2h(); h2h(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res['host_to_host'] = {'value': round(2 * B / dt, 1), 'unit': 'proofs/s', 'note': 'input blocks (334 x 32 B per voter) in pinned host memory when the clock starts, proofs and public '
                           'signals in host memory when it stops; 2 steps of %d' % B}
    # ---- [r5] census to proofs: the voters' RAW data (address, password, signature, weight) on the host when the clock starts -- SIKs, nullifiers, both trees and every sibling
    #      list (zkc_census_inputs, on the GPU, straight into device-resident input blocks), witnesses, proofs -- proofs of B of them on the host when it stops ----
    nC = max(8192, B); eid, address, password, signature, avail = census._voter_data(nC, census.ELECTION_ID_HEX)
    vh = [census.bytes_to_arbo(a.to_bytes((a.bit_length() + 7) // 8 or 1, 'big')) for a in avail]
    d_all = torch.empty(nC * len(flat) // B, dtype=torch.uint8, device='cuda')

    def c2p():
        census.census_inputs(ctx, eid, address, password, signature, avail, [1] * nC, vh, nl, d_all.data_ptr())
        return pk.fullprove_batch_dev(d_all.data_ptr(), B, d_wtns.data_ptr(), d_status.data_ptr(), draw_rs(rs, 2 * B).tobytes())
    c2p(); torch.cuda.synchronize(); t0 = time.perf_counter(); pc, uc = c2p(); torch.cuda.synchronize(); dtc = time.perf_counter() - t0
    t0 = time.perf_counter(); census.census_inputs(ctx, eid, address, password, signature, avail, [1] * nC, vh, nl, d_all.data_ptr()); dtb = time.perf_counter() - t0
    res['census_to_proofs'] = {'census_voters': nC, 'proved': B, 'seconds': round(dtc, 4), 'census_build_s': round(dtb, 4), 'proofs_per_s_census_build_included': round(B / dtc, 1),
                               'public_signals_equal_the_timed_step': bool(uc == out['pubs']),
                               'note': 'the whole %d-voter census is built (native builder: trie split in C++, hashes and sibling scatter on the GPU; the Python builder of rounds 1-4 '
                                       'took ~10 s) and the first %d voters proved, one call each; the Python side of the builder call (ints -> bytes) is inside census_build_s' % (nC, B)}
    del d_all
    # ---- one proof (BASELINE configs[1]): the first voter of the batch alone, inputs -> proof in one call, inputs and outputs' device buffers as in the headline ----
    one = []
    for _ in range(14):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p1, u1 = pk.fullprove_batch_dev(d_inputs.data_ptr(), 1, d_wtns.data_ptr(), d_status.data_ptr(), out['rs'][:64])
        one.append((time.perf_counter() - t0) * 1e3)
    one = sorted(one[2:])
    res['single_proof'] = {'ms_min': round(one[0], 3), 'ms_median': round(one[len(one) // 2], 3), 'bytes_equal_proof_of_the_timed_batch': bool(p1 == out['proofs'][:256] and u1 == out['pubs'][:len(u1)]),
                           'note': 'voter 0 of the timed batch alone, same (r, s): a pass of one proof takes the latency-shaped path (blinding without variable-base products, small virtual '
                                   'windows, half a wave per G2 bucket; DESIGN.md section 7) and gives the bytes the batch gave; 12 calls after 2 warm-ups.  The witness chain grows with the depth of the voter\'s leaf: this census puts it 13-14 levels down '
                                   '(the reference\'s inputs_example.json, shallower, takes 3.2 ms: profiles/r03_single_proof_latency.json)'}
    # ---- folding: worst cases ----
    Bd = min(B, 188)
    deep = census.deep_voters(ctx, Bd, nl)
    dflat = b''.join(zkcensus_amd.flatten_inputs(v, nl) for v in deep)
    dd_in = torch.from_numpy(np.frombuffer(dflat, dtype=np.uint8).copy()).cuda()
    dd_w = torch.empty(Bd * nW * 32, dtype=torch.uint8, device='cuda'); dd_st = torch.zeros(Bd, dtype=torch.int32, device='cuda')
    rsd = draw_rs(rs, 2 * Bd).tobytes()
    pk.fullprove_batch_dev(dd_in.data_ptr(), Bd, dd_w.data_ptr(), dd_st.data_ptr(), rsd)
    assert int(dd_st.abs().sum().item()) == 0
    ctx._lib.zkc_profile_enable(ctx._h, 0x10); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        pd, ud = pk.fullprove_batch_dev(dd_in.data_ptr(), Bd, dd_w.data_ptr(), dd_st.data_ptr(), rsd)
    torch.cuda.synchronize(); dt_deep = time.perf_counter() - t0
    madds_deep = read_prof(ctx)['msm_g1_streamed']['launches'] / (2 * Bd); ctx._lib.zkc_profile_enable(ctx._h, 0)
    ok_deep = groth16.verify_batch(ctx, vk, ud, pd)
    # the same key without the folding tables, witness given (no witness generation in this leg): the foreign-witness path of groth16.prove(zkey, wtns)
    Bu = min(B, 188)
    os.environ['ZKC_NO_FOLD'] = '1'
    try:
        pk_nf = zkcensus_amd.ProvingKey(ctx, zkey_bytes)
    finally:
        del os.environ['ZKC_NO_FOLD']
    rsu = out['rs'][:64 * Bu]
    pu, uu = pk_nf.prove_batch_dev(d_wtns.data_ptr(), Bu, rsu)
    same = pu == out['proofs'][:256 * Bu] and uu == out['pubs'][:256 * Bu]
    ctx._lib.zkc_profile_enable(ctx._h, 0x10); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(2):
        pk_nf.prove_batch_dev(d_wtns.data_ptr(), Bu, rsu)
    torch.cuda.synchronize(); dt_nf = time.perf_counter() - t0
    madds_nf = read_prof(ctx)['msm_g1_streamed']['launches'] / (2 * Bu); ctx._lib.zkc_profile_enable(ctx._h, 0)
    pk_nf.close()
    res['folding'] = {'enabled': os.environ.get('ZKC_NO_FOLD') is None, 'madds_per_proof': madds_folded, 'madds_unfolded': round(madds_nf),
                      'unfolded_proofs_per_s': round(2 * Bu / dt_nf, 1), 'unfolded_bytes_equal_folded': bool(same),
                      'unfolded_note': 'key loaded with ZKC_NO_FOLD=1, %d witnesses of the timed step given (groth16.prove(zkey, wtns) shape: no witness generation in this leg), '
                                       'same (r, s): proofs byte-identical to the folded ones of the timed step' % Bu,
                      'depth160_proofs_per_s': round(2 * Bd / dt_deep, 1), 'depth160_madds_per_proof': round(madds_deep), 'depth160_all_valid': bool(ok_deep),
                      'depth160_note': '%d voters whose leaves sit %d levels down both trees (every sibling non-zero), inputs -> witness -> proof through the SAME key as the headline: folding '
                                       'stays enabled and removes only the old-key block; such passes run their sections over the key\'s second (15-bit) tables (DESIGN.md section 3)' % (Bd, nl)}
    del dd_in, dd_w, dd_st
    # ---- isolated stage times: one pass, one stream ----
    Bs = min(B, int(os.environ.get('ZKC_INFLIGHT', '96')) - 2)
    os.environ['ZKC_SERIAL_STREAMS'] = '1'
    try:
        pk_s = zkcensus_amd.ProvingKey(ctx, zkey_bytes)
    finally:
        del os.environ['ZKC_SERIAL_STREAMS']
    rss = out['rs'][:64 * Bs]
    ps, us = pk_s.fullprove_batch_dev(d_inputs.data_ptr(), Bs, d_wtns.data_ptr(), d_status.data_ptr(), rss)
    same_s = ps == out['proofs'][:256 * Bs]
    ctx._lib.zkc_profile_enable(ctx._h, 0x7f); torch.cuda.synchronize(); t0 = time.perf_counter()
    pk_s.fullprove_batch_dev(d_inputs.data_ptr(), Bs, d_wtns.data_ptr(), d_status.data_ptr(), rss)
    torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
    prof = read_prof(ctx); ctx._lib.zkc_profile_enable(ctx._h, 0)
    pk_s.close()
    stages = {}
    for k, v in prof.items():
        if k == 'msm_g1_streamed' or v['ms'] <= 0:
            continue
        gbps = v['alg_bytes'] / (v['ms'] * 1e-3) / 1e9
        stages[k] = {'ms_isolated': round(v['ms'], 3), 'alg_bytes': v['alg_bytes'], 'GBps': round(gbps, 1), 'frac': round(gbps / HBM_PEAK_GBPS, 4)}
    res['stages_isolated'] = {'proofs_in_pass': Bs, 'pass_ms_serial': round(dt_s * 1e3, 2), 'sum_of_stage_ms': round(sum(v['ms_isolated'] for v in stages.values()), 2),
                              'bytes_equal_pipelined_proofs': bool(same_s), 'stages': stages,
                              'note': 'one pass of %d proofs with every kernel on one stream (ZKC_SERIAL_STREAMS=1): isolated, additive stage times; alg_bytes per SURVEY.md 8d '
                                      '(whole sections for the MSMs, 6 transforms + joinABC for ntt_joinABC); frac = GB/s over 8000.  The pipelined pass overlaps these on three streams' % Bs}
    return res


def _le32(x):
    return int(x).to_bytes(32, 'little')


def service_leg(args, ctx, dev, zkey_bytes, vk, voters, flat, d_wtns, nW, nIn):
    """[r5] The reference's own call shape: ONE voter per call -- prover.Prove(zkey, wasm, inputs) per voter from goroutines (zk_census_test.go:89, ending in
    rapidsnark's groth16_prover) and groth16.fullProve per ballot (ts_inputs/src/example.ts:358-362) -- from 64 and 256 concurrent callers over voters of the timed batch.
    The callers are native threads (tools/loadgen/loadgen.c; Python threads would put the interpreter lock between them); all end in the library's proving service
    (csrc/zkc_service.hip), which forms pipeline passes out of whoever is waiting.  Every proof of every leg goes through the batch verifier."""
    from zkcensus_amd import groth16
    lib = ctx._lib
    so = os.path.join(ROOT, 'tools', 'loadgen', 'libzkc_loadgen.so')
    if not os.path.exists(so):
        return {'skipped': 'tools/loadgen/libzkc_loadgen.so is not built (python -c "import __graft_entry__ as g; g.build()")'}
    lg = ctypes.CDLL(so); vp = ctypes.c_void_p
    lg.zkc_loadgen_run.argtypes = [ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                   ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    os.environ.setdefault('ZKC_DEVICE', str(dev))                       # the process-wide service behind groth16_prover stays on this rank's GPU
    V = min(256, len(voters))
    wt = d_wtns.view(-1, nW * 32)[:V].cpu().numpy()
    images = []
    for i in range(V):                                                  # .wtns file images: what go-rapidsnark hands to groth16_prover
        w = wt[i].tobytes(); n = lib.zkc_wtns_write(w, nW, None, 0); buf = ctypes.create_string_buffer(n); lib.zkc_wtns_write(w, nW, buf, n); images.append(buf.raw)
    flats = [flat[i * nIn * 32:(i + 1) * nIn * 32] for i in range(V)]
    texts = [json.dumps(voters[i]).encode() for i in range(V)]          # inputs_example.json's shape: what prover.Prove receives (zk_census_test.go:85-89)
    h = lib.zkc_service_default()
    if not h:
        return {'skipped': (lib.zkc_service_last_error() or b'').decode()}
    svc = ctypes.c_void_p(h)

    def stats():
        a = (ctypes.c_uint64 * 8)(); b = (ctypes.c_uint64 * 8)(); lib.zkc_service_stats(svc, a); lib.zkc_service_timing(svc, b)
        return [int(x) for x in a], [int(x) for x in b]

    def leg(mode, callers, calls):
        n = callers * calls
        items = {0: images, 1: flats, 4: texts}[mode]
        arr = (ctypes.c_char_p * V)(*items); lens = (ctypes.c_size_t * V)(*[len(x) for x in items])
        wall = ctypes.c_double(0); lat = (ctypes.c_double * n)()
        pj = uj = pr = pu = st = None
        if mode == 1:
            pr = ctypes.create_string_buffer(256 * n); pu = ctypes.create_string_buffer(256 * n); st = (ctypes.c_int32 * n)()
            fn = ctypes.cast(lib.zkc_service_fullprove, vp)
        else:
            pj = ctypes.create_string_buffer(2048 * n); uj = ctypes.create_string_buffer(2048 * n)
            fn = ctypes.cast(lib.groth16_prover if mode == 0 else lib.groth16_fullprove, vp)
        (s0, t0) = stats(); c0 = time.process_time()
        failed = lg.zkc_loadgen_run(mode, fn, svc if mode == 1 else None, callers, calls, zkey_bytes, len(zkey_bytes), None, 0, args.nlevels, 8, arr, lens, V, pj, uj, pr, pu, st,
                                    ctypes.byref(wall), lat)
        cpu_s = time.process_time() - c0; (s1, t1) = stats()
        if mode == 1:
            proofs, pubs = pr.raw, pu.raw
        else:                                                           # JSON texts, parsed after the clock has stopped
            pl, ul = [], []
            for i in range(n):
                p = json.loads(pj.raw[2048 * i:2048 * (i + 1)].split(b'\0', 1)[0]); u = json.loads(uj.raw[2048 * i:2048 * (i + 1)].split(b'\0', 1)[0])
                pl += [_le32(x) for x in (p['pi_a'][0], p['pi_a'][1], p['pi_b'][0][0], p['pi_b'][0][1], p['pi_b'][1][0], p['pi_b'][1][1], p['pi_c'][0], p['pi_c'][1])]
                ul += [_le32(x) for x in u]
            proofs, pubs = b''.join(pl), b''.join(ul)
        ok = failed == 0 and groth16.verify_batch(ctx, vk, pubs, proofs)
        nb = max(1, s1[1] - s0[1]); ls = sorted(lat)
        return {'callers': callers, 'calls_per_caller': calls, 'proofs': n, 'seconds': round(wall.value, 4), 'proofs_per_s': round(n / wall.value, 1), 'all_verified_by_batch_verifier': bool(ok),
                'batches': nb, 'mean_batch': round(n / nb, 1), 'largest_batch_so_far': s1[2], 'latency_ms_p50': round(ls[n // 2], 2), 'latency_ms_p95': round(ls[int(n * 0.95)], 2),
                'worker_ms_per_batch': {k: round((t1[i] - t0[i]) / 1e3 / nb, 2) for i, k in enumerate(('upload', 'collect', 'key', 'begin_to_finish', 'hand_back'))},
                'host_cpu_cores_used': round(cpu_s / wall.value, 2)}
    res = {'what': 'one voter per call from N concurrent native caller threads through the proving service; callers x calls proofs per leg, every one batch-verified; '
                   'groth16_prover takes whole .zkey / .wtns file images and returns JSON (the symbol go-rapidsnark binds), zkc_service_fullprove takes the 334 x 32-byte input block, '
                   'groth16_fullprove the byte slices of prover.Prove(zkey, wasm, inputs) with the inputs as JSON text (wasm NULL here: the circuit by the key\'s shape)'}
    leg(0, 8, 1); leg(1, 64, 2); leg(0, 64, 2); leg(0, 256, 1); leg(1, 256, 1); leg(4, 8, 1)     # untimed: key load, work space, staging buffers, pinned slots
    res['groth16_prover'] = [leg(0, 64, 32), leg(0, 256, 12)]
    res['zkc_service_fullprove'] = [leg(1, 64, 32), leg(1, 256, 12)]
    res['groth16_fullprove_json'] = [leg(4, 64, 32)]
    res['lone_sequential_caller_ms'] = leg(1, 1, 20)['latency_ms_p50']      # one caller, one call at a time: the latency through the same queue
    mem = (ctypes.c_uint64 * 8)(); lib.zkc_service_memory(svc, mem)
    res['memory_GB'] = dict(zip(('resident_keys', 'key_tables', 'lanes_work_space_shared_by_the_keys', 'largest_key_tables', 'largest_device_work_space', 'staging_device', 'pinned_host', 'reserve_failures'),
                                [int(mem[0])] + [round(int(x) / 1e9, 2) for x in mem[1:7]] + [int(mem[7])]))
    return res


def generic_2p20_leg(ctx):
    """[r5] BASELINE configs[4] where the driver sees it: full Groth16 proofs of a circuit-shaped random R1CS at the 2^20 ceiling of the reference's powers of
    tau (circuit/circuit-compiler.sh:57) -- 983 040 constraints, 983 105 wires, four different witnesses -- through groth16.prove's generic path; five of them through the pinned
    verifier (the closed-form comparison of this size is in tools/generic_bench.py: 15 s of Python)."""
    import random, tempfile, shutil
    import numpy as np, torch
    import zkcensus_amd
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib as ol, big_circuit as bc                          # instance generator and the checker
    from test_generic_circuit import setup_key
    logn = 20; n = 1 << logn; n_cons = n - n // 16; n_in = 64; n_pub = 8; B = 24; NW = 4
    tmp = tempfile.mkdtemp(prefix='zkc_bench_2p20_')
    try:
        t0 = time.time(); r1 = os.path.join(tmp, 'c.r1cs'); n_wires = bc.chain_instance(r1, n_cons, n_in, n_pub, seed=logn)
        rng = random.Random(logn)
        wits = [bc.chain_witness(n_cons, n_in, logn, [rng.getrandbits(253) for _ in range(n_in)]) for _ in range(NW)]
        zk, vk = setup_key(r1, 2024 + logn); t_host = time.time() - t0
        t0 = time.time(); pk = zkcensus_amd.ProvingKey(ctx, zk); torch.cuda.synchronize(); t_load = time.time() - t0
        d_w = torch.from_numpy(np.frombuffer(b''.join(wits), dtype=np.uint8).copy()).cuda().repeat((B + NW - 1) // NW)[:B * len(wits[0])].contiguous()
        rs = b''.join(_le32(3 + 2 * k) + _le32(5 + 3 * k) for k in range(B))
        proofs, pubs = pk.prove_batch_dev(d_w.data_ptr(), B, rs)        # warm-up: the work space grows here
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2):
            proofs, pubs = pk.prove_batch_dev(d_w.data_ptr(), B, rs)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        npb = len(pubs) // B
        ok = all(ol.verify(vk, pubs[npb * i:npb * (i + 1)], proofs[256 * i:256 * i + 256]) for i in (0, 1, 2, 3, B - 1))
        t1 = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); pk.prove_batch_dev(d_w.data_ptr(), 1, rs[:64]); t1.append((time.perf_counter() - t0) * 1e3)
        pk.close(); del d_w; torch.cuda.empty_cache()
        return {'domain': n, 'constraints': n_cons, 'wires': n_wires, 'zkey_MB': round(len(zk) / 1e6, 1), 'batch': B, 'generic_2p20_proofs_per_s': round(B / dt, 2), 'ms_per_proof': round(dt / B * 1e3, 2),
                'lone_proof_ms': round(min(t1), 2), 'oracle_verifier_accepts_sampled': bool(ok), 'host_seconds_instance_and_setup': round(t_host, 1), 'key_load_s': round(t_load, 2),
                'what': 'groth16.prove(zkey, wtns) for a circuit that is not the census circuit: no witness generation, no constant folding; witnesses device-resident, four different ones tiled over the batch'}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1')); local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.dry_run_cpu:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    import torch.distributed as dist
    torch.set_num_threads(1)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    # ZKC_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box only): ranks share the visible devices and gather over gloo, because RCCL refuses two
    # ranks on one device.  Never set for a measurement.
    share = os.environ.get('ZKC_BENCH_SHARE_GPU') == '1'
    ndev = torch.cuda.device_count()
    isolated = ndev == 1 and any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
    if share:
        dev = local % ndev
    elif local < ndev:
        dev = local
    elif isolated:
        dev = 0                    # a launcher that hands every rank its own card through *_VISIBLE_DEVICES
    else:
        raise SystemExit('bench.py: local rank %d but only %d visible GPU(s); one rank per GPU (ZKC_BENCH_SHARE_GPU=1 is the one-GPU rehearsal)' % (local, ndev))
    torch.cuda.set_device(dev)
    if world > 1:
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev))

    import zkcensus_amd
    from zkcensus_amd import setup, census, parallel, groth16
    # ---- artifacts: test proving key (the reference's proving_key.zkey is a missing blob) ----
    if local == 0:
        setup.ensure_test_artifacts(args.nlevels)
    if world > 1:
        dist.barrier()
    _, zkey_path, vkey_path = setup.ensure_test_artifacts(args.nlevels)
    ctx = zkcensus_amd.Context(dev)
    zkey_bytes = open(zkey_path, 'rb').read()
    pk = zkcensus_amd.ProvingKey(ctx, zkey_bytes)
    B = args.batch
    # ---- synthetic census: B voters per rank, this rank proves block `rank` ----
    # the census is the 8 192-voter one of configs 3/4 whatever N is (leaf depth 13-17 decides how much of a witness folds away);
    # rank r proves voters [r B, (r+1) B)
    lo, hi = parallel.shard_range(rank, world, B * world)
    # [r4] the census is built ONCE, by rank 0, and every rank is handed its block of input records over the process group (one broadcast of total x 334 x 32 bytes: 87 MB
    # for 8 192 voters) -- not eight identical builds beside each other.  ZKC_BENCH_CENSUS_PER_RANK=1: the old form.
    # [r5] ... by the NATIVE builder (zkc_census_inputs, csrc/zkc_census.hip: trie split in C++, every hash and the sibling scatter on the GPU): ~0.1 s where the Python
    # builder took 10.  The 12-key objects the checkers want are made on demand from the flat blocks (census.FlatVoters).
    nIn = ctx.n_inputs(args.nlevels)
    t_census = time.perf_counter()
    if world == 1 or os.environ.get('ZKC_BENCH_CENSUS_PER_RANK') == '1':
        allb, _, _ = census.synthetic_census_flat(ctx, max(8192, B * world), args.nlevels)
        flat = allb[lo * nIn * 32:hi * nIn * 32]; del allb
        voters = census.FlatVoters(flat, args.nlevels)
        d_inputs = torch.from_numpy(np.frombuffer(flat, dtype=np.uint8).copy()).cuda(dev)
    else:
        voters = None
        if rank == 0:
            allb, _, _ = census.synthetic_census_flat(ctx, max(8192, B * world), args.nlevels)
            allflat = torch.from_numpy(np.frombuffer(allb[:B * world * nIn * 32], dtype=np.uint8).copy())
            voters = census.FlatVoters(allb[lo * nIn * 32:hi * nIn * 32], args.nlevels); del allb
        else:
            allflat = torch.empty(B * world * nIn * 32, dtype=torch.uint8)
        d_all = allflat if share else allflat.cuda(dev)
        dist.broadcast(d_all, src=0)
        d_inputs = d_all[lo * nIn * 32:hi * nIn * 32].clone().cuda(dev)
        flat = bytes(d_inputs.cpu().numpy().tobytes())
        del d_all, allflat
    t_census = time.perf_counter() - t_census
    nW = ctx.n_wires(args.nlevels)
    # two sets of witness / status buffers: a step is begun (everything enqueued: zkc_batch_begin) before the previous one is finished (zkc_batch_finish), so that
    # the tail of step k -- bucket reduction and blinding of its last pass, copies -- runs beside the head of step k + 1 (its first witness kernels and transforms):
    # what a caller that keeps the GPU fed does (the proving service does the same for per-voter callers).  Every step is still begun AND finished inside the timed
    # region; --no-step-pipelining makes each step one synchronous zkc_fullprove_batch_dev call as in rounds 1 and 2.
    pipelined = not args.no_step_pipelining
    d_wtns_s = [torch.empty(B * nW * 32, dtype=torch.uint8, device='cuda') for _ in range(2 if pipelined else 1)]
    d_status_s = [torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(len(d_wtns_s))]
    out = {}
    rs = np.random.default_rng(0x5A4B43454E535553 + rank)
    state = {'prev': None, 'i': 0}

    def collect(slot, rsb, p, pub):
        rec = parallel.pack_records(p, pub, d_status_s[slot].cpu().tolist())      # 256 B proof + 8 x 32 B signals + status per voter
        # RCCL over xGMI: the only collective -- finished proofs to every rank (513 B per voter)
        if world > 1:
            out['records'] = parallel.gather_records(rec if share else rec.cuda(dev), world, dist, B * world)
        else:
            out['records'] = rec
        out['rec'], out['proofs'], out['pubs'], out['rs'], out['slot'] = rec, p, pub, rsb, slot

    def step():
        rsb = draw_rs(rs, 2 * B).tobytes()                                    # (r, s) per proof, uniform in Fr
        # inputs -> witness -> proof for the whole batch (groth16.fullProve per voter, ts_inputs/src/example.ts:358) through the C ABI
        if not pipelined:
            p, pub = pk.fullprove_batch_dev(d_inputs.data_ptr(), B, d_wtns_s[0].data_ptr(), d_status_s[0].data_ptr(), rsb)
            return collect(0, rsb, p, pub)
        slot = state['i'] & 1; state['i'] += 1
        pk.batch_begin(slot, d_inputs.data_ptr(), B, d_wtns_s[slot].data_ptr(), d_status_s[slot].data_ptr(), rsb)
        drain()
        state['prev'] = (slot, rsb)

    def drain():
        if state['prev'] is not None:
            slot, rsb = state['prev']; state['prev'] = None
            p, pub = pk.batch_finish(slot, B)
            collect(slot, rsb, p, pub)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    if args.warmup:
        assert all(int(t.abs().sum().item()) == 0 for t in d_status_s), 'a synthetic voter failed a circuit assert'
    ctx._lib.zkc_profile_enable(ctx._h, 0x7f)
    def thread_cpu():                                        # per-thread CPU seconds of this process (diagnostics: ZKC_BENCH_THREAD_CPU=1 prints who used the host inside the timed region)
        res = {}
        try:
            tck = os.sysconf('SC_CLK_TCK')
            for tid in os.listdir('/proc/self/task'):
                f = open('/proc/self/task/%s/stat' % tid).read(); name = f[f.index('(') + 1:f.rindex(')')]; v = f[f.rindex(')') + 2:].split()
                res[tid] = (name, (int(v[11]) + int(v[12])) / tck)
        except OSError:
            pass
        return res
    sync()
    tc0 = thread_cpu() if os.environ.get('ZKC_BENCH_THREAD_CPU') else None
    t0 = time.perf_counter(); c0 = time.process_time()
    for _ in range(args.steps):
        step()
    drain()
    sync()
    dt = time.perf_counter() - t0
    # [r5] host CPU this rank spent per second of wall time inside the timed region (all threads of the process: the enqueueing thread, the HIP runtime's, RCCL's proxies).
    # The GPU boxes hand a container 16 cores' worth of CPU time (cpu.max; profiles/r04_cpu_baseline_scaling.json): eight ranks must fit in that.
    cpu_used = (time.process_time() - c0) / dt
    if tc0 is not None:
        tc1 = thread_cpu()
        sys.stderr.write('bench.py: host CPU per thread inside the timed region (%.2f s wall): %s\n' % (dt, sorted(((round(c - tc0.get(t, (n, 0))[1], 2), n, t) for t, (n, c) in tc1.items()), reverse=True)[:10]))
    d_wtns, d_status = d_wtns_s[out['slot']], d_status_s[out['slot']]          # buffers of the LAST timed step: what the verification legs below look at
    tmax = torch.tensor([dt], dtype=torch.float64, device='cuda')
    if world > 1 and not share:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elif world > 1:
        t = tmax.cpu(); dist.all_reduce(t, op=dist.ReduceOp.MAX); tmax = t
    per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    if world > 1:                                         # every rank's own step time beside the maximum: a straggler is visible in the line
        mine = torch.tensor([dt], dtype=torch.float64) if share else torch.tensor([dt], dtype=torch.float64, device='cuda')
        each = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(each, mine)
        per_rank_ms = [round(float(t.item()) / args.steps * 1e3, 3) for t in each]
    cpu_per_rank = [round(cpu_used, 3)]
    if world > 1:
        mine = torch.tensor([cpu_used], dtype=torch.float64) if share else torch.tensor([cpu_used], dtype=torch.float64, device='cuda')
        each = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(each, mine)
        cpu_per_rank = [round(float(t.item()), 3) for t in each]
    dt = float(tmax.item())
    assert all(int(t.abs().sum().item()) == 0 for t in d_status_s), 'a synthetic voter failed a circuit assert'

    # ---- per-category device time from HIP events on the library's stream ----
    prof = read_prof(ctx)
    ctx._lib.zkc_profile_enable(ctx._h, 0)
    # the kernel BASELINE.json's metric names ("MSM HBM GB/s vs peak") and the one that moves most algorithmic bytes: G1 bucket
    # accumulation.  (Per-category times overlap across the three streams, so 'largest time' is not a reliable selector.)
    dom = 'msm_accumulate_g1'
    d = prof[dom]
    achieved = d['alg_bytes'] / (d['ms'] * 1e-3) / 1e9 if d['ms'] > 0 else 0.0
    traffic, traffic_src = None, None
    for pmc_file in ('r05_pmc_hbm_traffic.json', 'r04_pmc_hbm_traffic.json', 'r03_pmc_hbm_traffic.json', 'r02_pmc_hbm_traffic.json', 'r01_pmc_hbm_traffic.json'):
        try:      # HBM bytes per launch of the same kernel/geometry from the committed rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
            pm = json.load(open(os.path.join(ROOT, 'profiles', pmc_file)))
            traffic = int(pm.get('hbm_bytes_per_launch', pm['hbm_bytes_per_launch_uncorrected']))
            traffic_src = ('profiles/%s (separate --pmc FETCH_SIZE / WRITE_SIZE runs, %d proofs per launch; FETCH_SIZE x %.2f: the factor the guide asks to calibrate for one\'s own access pattern, '
                           'measured for random 64-byte gathers in profiles/r02_pmc_fetch_calibration.json -- with the guide\'s factor 2 for wide coalesced streams it would be %.1f GB)'
                           % (pmc_file, pm['proofs_per_launch'], pm.get('fetch_calibration_factor', 1.0), pm.get('hbm_bytes_per_launch_with_stream_factor_2', 0) / 1e9))
            break
        except Exception:
            pass
    # the bound that actually holds: VALU issue.  profiles/r02_rate_probe.txt (tools/probe/rate_probe.hip) holds the measured lane-instruction
    # rates of this part; profiles/r02_accumulate_isa_histogram.txt the instruction mix of one mixed addition of the kernel; VALU_PROFILE below is
    # read from profiles/r02_valu_model.json when present (written by tools/valu_model.py from those two files) so that the figure is reproducible
    # from committed files.
    vm = {'mad_u64_u32_per_madd': 1476, 'instr_per_madd': 2400, 'rate_mad_u64_u32': 34.8e12, 'rate_valu32': 57.2e12, 'source': 'round-1 constants (no committed probe output)'}
    try:
        vm = json.load(open(os.path.join(ROOT, 'profiles', 'r03_valu_model.json' if os.path.exists(os.path.join(ROOT, 'profiles', 'r03_valu_model.json')) else 'r02_valu_model.json')))
    except Exception:
        pass
    nproofs = args.steps * B
    madds = prof['msm_g1_streamed']['launches']                                          # counted on the device: non-zero signed digits = mixed additions performed
    madd_rate = madds / (d['ms'] * 1e-3) if d['ms'] > 0 else 0.0
    capacity = vm.get('capacity_madd_per_s') or 1.0 / (vm['mad_u64_u32_per_madd'] / vm['rate_mad_u64_u32'] + (vm['instr_per_madd'] - vm['mad_u64_u32_per_madd']) / vm['rate_valu32'])
    alu = {'unit': 'mixed additions/s', 'achieved': round(madd_rate / 1e9, 3), 'achieved_unit': 'G madd/s', 'madds_per_proof': round(madds / max(1, nproofs)), 'instr_per_madd': vm['instr_per_madd'],
           'mad_u64_u32_per_madd': vm['mad_u64_u32_per_madd'], 'peak': round(capacity / 1e9, 2), 'peak_unit': 'G madd/s', 'frac': round(madd_rate / capacity, 4),
           'model_source': vm.get('source'),
           'note': 'VALU issue bound: capacity = lane-operations/s a dependency-free loop with the kernel\'s own instruction mix sustains on this part '
                   '(tools/probe/rate_probe.hip, profiles/r02_rate_probe.txt) / VALU instructions of one mixed addition (profiles/r02_accumulate_isa_histogram.txt); '
                   'rocprofv3 SQ counters of the kernel are in profiles/r02_pmc_sq_accumulate.json.  This, not HBM, limits the kernel; box-to-box clock differences move frac by a few percent'}
    # [r5] the whole pipeline against the chip's vector issue: VALU-busy cycles per SIMD of ONE pass, summed over every kernel (rocprofv3 SQ counters, committed:
    # profiles/r05_kernel_valu_busy_table.json), per proof, times the measured proofs/s, over the clock the accumulation kernel sustains (2.15 GHz, profiles/r02_power_clock_trace.json)
    pipeline_valu = None
    try:
        kb = json.load(open(os.path.join(ROOT, 'profiles', 'r05_kernel_valu_busy_table.json')))
        per_proof = kb['pass_valu_busy_Mcycles_per_simd'] * 1e6 / kb.get('proofs_in_pass', 64)
        pipeline_valu = {'valu_busy_cycles_per_simd_per_proof': round(per_proof), 'proofs_per_s_if_every_issue_slot_were_used_at_2.15_GHz': round(2.15e9 / per_proof, 1),
                         'frac_of_issue_slots_used': round(args.steps * B / dt * per_proof / 2.15e9, 4), 'source': 'profiles/r05_kernel_valu_busy_table.json (one pass, every kernel, additive)',
                         'note': 'at N = 1; what is left of the chip is this fraction short of 1: the pipeline as a whole, not one kernel, is at the vector-issue ceiling'}
    except Exception:
        pass
    roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'valu': alu,
                'frac': round(achieved / HBM_PEAK_GBPS, 6), 'pipeline_valu': pipeline_valu, 'traffic': traffic, 'traffic_source': traffic_src,
                'avg_launch_ms': round(d['ms'] / max(1, d['launches']), 4), 'alg_bytes_per_launch': d['alg_bytes'] // max(1, d['launches']),
                'streamed_pair_bytes_per_launch': prof['msm_g1_streamed']['alg_bytes'] // max(1, d['launches']),
                'note': 'achieved = algorithmic bytes (whole A,B1,C,H sections, SURVEY.md 8d) / kernel time; constant folding streams only streamed_pair_bytes. '
                        'The kernel is bound by VALU issue (see valu), not by HBM; traffic exceeds the algorithmic bytes because every (scalar, window) digit '
                        'gathers its own pre-shifted 64-byte base (15-22 table rows per base point): a deliberate deviation from coalesced streaming, see DESIGN.md'}

    # ---- verify what was timed: the proofs of the LAST timed step ----
    verified = None
    vk = json.load(open(vkey_path))
    inflight = pk.pass_size
    npass = -(-B // inflight); per = -(-B // npass)                       # the library cuts a batch into equal passes of at most `inflight` proofs (zkc_zkey_pass_info)
    sample = sorted({i for i in (0, 1, per - 1, per, 2 * per - 1, 2 * per, B // 2, (npass - 1) * per - 1, (npass - 1) * per, B - 2, B - 1) if 0 <= i < B})
    if not args.no_verify:
        groth16.verify_batch(ctx, vk, out['pubs'], out['proofs'])                            # once untimed: the key is made ready (its checks, prepared gamma / delta), the verifier's kernels load, its work space is allocated
        t_v = time.perf_counter()
        ok_batch = groth16.verify_batch(ctx, vk, out['pubs'], out['proofs'])                  # all B proofs of the step, product batch verifier (Miller loops on the GPU)
        batch_verify_ms = 1e3 * (time.perf_counter() - t_v)
        from zkcensus_amd import _native as _nat
        vkb = groth16.vk_to_bytes(vk); _vlib = _nat.load(); t_v = time.perf_counter()
        ok_single = [_vlib.zkc_verify_bin(vkb, 8, out['pubs'][256 * i:256 * i + 256], out['proofs'][256 * i:256 * i + 256]) == 1 for i in range(min(B, 16))]   # proof.Verify per vote (zk_census_test.go:122), CPU
        single_verify_ms = 1e3 * (time.perf_counter() - t_v) / len(ok_single)
        ok_batch = ok_batch and all(ok_single)
        ok_gather = bool((out['records'][lo:hi].cpu() == out['rec']).all())                  # gathered records carry this rank's proofs
        flags = torch.tensor([1 if ok_batch else 0, 1 if ok_gather else 0])
        if world > 1:
            f = flags if share else flags.cuda(dev)
            dist.all_reduce(f, op=dist.ReduceOp.MIN); flags = f.cpu()
        verified = {'step': 'last timed step', 'proofs': B * world, 'batch_verifier_all_valid': bool(flags[0].item()),
                    'gathered_records_equal_per_rank_records': bool(flags[1].item()),
                    'batch_verifier_ms_per_rank': round(batch_verify_ms, 2), 'batch_verifier_proofs_per_s_per_rank': round(B / (batch_verify_ms / 1e3), 0),
                    'single_verify_cpu_ms': round(single_verify_ms, 2)}

    cpu = None
    if rank == 0 and not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import oracle_lib as ol                                          # the checker / baseline, never the product path
        ok = [ol.verify(vk, out['pubs'][256 * i:256 * i + 256], out['proofs'][256 * i:256 * i + 256]) for i in sample]
        verified['oracle_verifier_sampled'] = len(sample); verified['oracle_verifier_indices'] = sample
        verified['oracle_verifier_all_valid'] = all(ok)
        assert all(ok), 'the CPU oracle verifier rejects a timed proof: %r' % [i for i, o in zip(sample, ok) if not o]
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_verify:
        # CPU baseline = the oracle's witness + Groth16 prove, one proof per host thread (ctypes releases the GIL), on voters of the timed batch
        # with the (r, s) the GPU used for them: each oracle proof doubles as a byte-for-byte parity check of a proof that was timed.
        from concurrent.futures import ThreadPoolExecutor
        # [r4] "all host cores" = the cores' worth of CPU TIME this container receives, which on the GPU boxes is far below the 256 cores the affinity mask shows: a CPU-time
        # quota (tools/cpu_baseline_scaling.py, profiles/r04_cpu_baseline_scaling.json: 64 threads receive the same ~16-32 cores' worth as 32, proofs/s flat, every proof slower).
        # cores_received() measures it in 0.4 s (the oracle's NTT on one thread per schedulable core: CPU seconds / wall); the leg runs on that many threads.
        avail = len(os.sched_getaffinity(0))
        quota = cores_received(ol, avail)
        try:                                                      # where the container shows its cgroup-v2 CPU quota (the GPU boxes: "1600000 100000" = 16 cores), that is the number
            q_us, per_us = open('/sys/fs/cgroup/cpu.max').read().split()
            if q_us != 'max':
                quota = min(quota, int(q_us) / int(per_us))
        except (OSError, ValueError):
            pass
        want = os.environ.get('ZKC_CPU_BASELINE_THREADS')
        cores = max(1, min(avail, int(want))) if want else max(1, min(avail, int(quota + 0.5)))
        wt = d_wtns.view(B, nW * 32)
        order = sample + [i for i in range(B) if i not in set(sample)]
        ol.lib()

        def one(i):
            rc, w = ol.witness(voters[i], args.nlevels); assert rc == 0
            r = int.from_bytes(out['rs'][64 * i:64 * i + 32], 'little'); s = int.from_bytes(out['rs'][64 * i + 32:64 * i + 64], 'little')
            rc, p, pub = ol.prove(zkey_bytes, w, r, s); assert rc == 0
            return i, w, p, pub
        t1 = time.perf_counter(); done = []
        with ThreadPoolExecutor(cores) as ex:
            k = 0; budget = float(os.environ.get('ZKC_CPU_BASELINE_SECONDS', '12')); said = t1     # a long budget turns this leg into a parity check of the whole batch
            while k < B and (not done or time.perf_counter() - t1 < budget):
                done += list(ex.map(one, order[k:k + cores])); k += cores
                if time.perf_counter() - said > 30:
                    said = time.perf_counter(); sys.stderr.write('bench.py: cpu_baseline / parity leg: %d oracle proofs so far\n' % len(done)); sys.stderr.flush()
        cdt = time.perf_counter() - t1
        for i, w, p, pub in done:
            assert bytes(wt[i].cpu().numpy().tobytes()) == w, 'GPU witness of voter %d differs from the CPU oracle' % i
            assert out['proofs'][256 * i:256 * i + 256] == p and out['pubs'][256 * i:256 * i + 256] == pub, 'GPU proof of voter %d differs from the CPU oracle' % i
        verified['oracle_prover_bytes_equal'] = len(done); verified['oracle_prover_indices'] = [i for i, *_ in done]
        cpu = {'value': round(len(done) / cdt, 4), 'unit': 'proofs/s', 'cores': cores, 'cores_available': avail, 'cores_on_host': os.cpu_count(), 'cores_received_when_all_are_asked_for': round(quota, 1), 'kind': 'port',
               'sample': '%d full proofs (witness + Groth16 prove) of voters of the timed batch, %d at a time on %d threads, %.1f s; '
                         'the build\'s own C oracle, not snarkjs/rapidsnark (neither can run here)' % (len(done), cores, cores, cdt)}

    extras = None; service = None; generic = None
    key_shape = {'nVars': pk.n_vars, 'domainSize': pk.domain_size, 'pass_size': pk.pass_size, 'lanes': pk.lanes}
    if rank == 0 and world == 1 and not args.no_extras and not args.no_verify:
        extras = extra_legs(args, ctx, pk, zkey_bytes, vk, B, d_inputs, d_wtns, d_status, flat, out, rs, roofline['valu']['madds_per_proof'])
        if os.environ.get('ZKC_BENCH_SERVICE', '1') != '0':
            service = service_leg(args, ctx, dev, zkey_bytes, vk, voters, flat, d_wtns, nW, nIn)
        if os.environ.get('ZKC_BENCH_GENERIC_2P20', '1') != '0':
            pk.close(); pk = None; torch.cuda.empty_cache()            # the census key's tables and work space are not needed any more
            generic = generic_2p20_leg(ctx)
    if rank == 0 and cpu is None:
        cpu = {'value': None, 'note': 'the CPU baseline (the C oracle on the host cores the container gets) is timed at N = 1 only, on rank 0, on a bounded sample: see the N = 1 line'
                                      if world > 1 else 'skipped (--no-cpu-baseline / --no-verify)'}
    if rank == 0:
        total = args.steps * B * world
        line = {
            'metric': 'zkCensus proofs/sec (nLevels=%d)' % args.nlevels, 'value': round(total / dt, 3), 'unit': 'proofs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'ms_per_step_per_rank': per_rank_ms,
            'host_cpu_cores_used': max(cpu_per_rank), 'host_cpu_cores_used_per_rank': cpu_per_rank,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u256 (8 x u32 Montgomery, BN254 Fr/Fq)',
            'data': 'synthetic',
            'config': {'workload': 'zkCensus nLevels=%d, batch of %d voter proofs per GPU per step (BASELINE configs[2]/[3] shape), '
                                   'own test zkey seed 0x5A4B43454E535553, synthetic %d-voter census' % (args.nlevels, B, max(8192, B * world)),
                       'step_pipelining': pipelined, 'batch_per_gpu': B, 'pass_size': key_shape['pass_size'], 'lanes': key_shape['lanes'], 'nVars': key_shape['nVars'], 'domainSize': key_shape['domainSize'], 'parallelism': 'independent proofs per GPU, RCCL all_gather of 513 B/proof'},
            'roofline': roofline, 'cpu_baseline': cpu, 'verified': verified,
            'service': service, 'generic_2p20': generic, 'census_build_s': round(t_census, 3), 'census_to_proofs': (extras or {}).get('census_to_proofs'),
            'host_to_host': (extras or {}).get('host_to_host'), 'single_proof': (extras or {}).get('single_proof'), 'folding': (extras or {}).get('folding'), 'stages_isolated': (extras or {}).get('stages_isolated'),
            'stage_ms_per_proof_overlapped_not_additive': {k: round(v['ms'] / (args.steps * B), 4) for k, v in prof.items() if k != 'msm_g1_streamed'},
        }
        if share:
            line['rehearsal'] = 'ranks share one GPU and gather over gloo (ZKC_BENCH_SHARE_GPU=1): control-path rehearsal, not a scaling measurement'
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if pk is not None:
        pk.close()
    ctx.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())
