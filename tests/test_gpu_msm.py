"""GPU parity of the HIP MSM (pcdhip_msm*, replaces ark-ec VariableBaseMSM::multi_scalar_mul reached from
/root/reference src/ec_cycle_pcd/mod.rs:171,179) against the CPU oracle and the golden vectors, through the
C-ABI.  Bar: bit-exact on affine coordinates (integer arithmetic; projective representatives are not unique)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GROUPS = [(c, g) for c in range(4) for g in (1, 2)]


def affine(co, cid, grp, xyz):
    return co.to_affine(cid, grp, xyz)


def check(co, ctx, cid, grp, pts, sc, inf=None, modes=(-1, 0, 3), offset=0, n=None):
    n = len(sc) if n is None else n
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts[offset:offset + n], sc[:n], inf=None if inf is None else inf[offset:offset + n], nthreads=8))
    for mode in modes:
        ctx.set_precompute(mode)
        b = ctx.bases_upload(cid, grp, pts, inf)
        got = co.to_affine(cid, grp, ctx.msm(b, sc[:n], offset=offset, n=n))
        b.free()
        assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), (cid, grp, mode, n)
    ctx.set_precompute(-1)


@pytest.mark.parametrize("cid,grp", GROUPS)
def test_golden(co, golden, gpu_ctx, cid, grp):
    g = golden("msm")
    pre = f"c{cid}_g{grp}_"
    bases, inf, sc = g[pre + "bases"], g[pre + "inf"], g[pre + "scalars"]
    for mode in (-1, 0, 2):
        gpu_ctx.set_precompute(mode)
        b = gpu_ctx.bases_upload(cid, grp, bases, inf)
        xy, oinf = gpu_ctx.to_affine(cid, grp, gpu_ctx.msm(b, sc))
        assert np.array_equal(xy[0], g[pre + "result_xy"]) and oinf[0] == g[pre + "result_inf"][0]
        ones = np.zeros_like(sc)
        ones[:, 0] = 1
        xy, oinf = gpu_ctx.to_affine(cid, grp, gpu_ctx.msm(b, ones))
        assert np.array_equal(xy[0], g[pre + "ones_xy"]) and oinf[0] == g[pre + "ones_inf"][0]
        xy, oinf = gpu_ctx.to_affine(cid, grp, gpu_ctx.msm(b, np.zeros_like(sc)))
        assert oinf[0] == 1 and not xy.any()
        b.free()
    gpu_ctx.set_precompute(-1)


@pytest.mark.parametrize("cid,grp,sizes", [(0, 1, (1, 2, 31, 32, 33, 1000, 1 << 14)), (1, 1, (33, 5000)), (0, 2, (33, 700)),
                                           (1, 2, (33, 500)), (2, 1, (33, 600)), (3, 1, (300,)), (2, 2, (120,)), (3, 2, (90,))])
@pytest.mark.parametrize("dist", [0, 1])
def test_vs_oracle(co, gpu_ctx, cid, grp, sizes, dist):
    fr = co.CURVE_FR[cid]
    for n in sizes:
        pts = co.gen_points(cid, grp, n, seed=11 + n)
        sc = co.gen_scalars(fr, n, seed=5 + n, dist=dist)
        check(co, gpu_ctx, cid, grp, pts, sc)


@pytest.mark.parametrize("cid,grp", GROUPS)
def test_ones_bucket_merge(co, gpu_ctx, cid, grp):
    """scalars equal to one go to a pseudo bucket that msm_merge_ones_kernel adds to bucket (window 0, digit 1): with scalars 1 and
    2^c + 1 both operands of that addition are finite points (the case the mailbox form of the Fq3-753 addition got wrong in that
    kernel: it computes in the plain form since); also only ones, ones next to an empty bucket 1, and a copy of the same point"""
    fr = co.CURVE_FR[cid]
    L = co.FIELD_N64[fr]
    pts = co.gen_points(cid, grp, 64, seed=171)
    for c in (8, 11):
        for other in (2, (1 << c) + 1, (1 << c) + 2, 1):
            for n in (2, 8, 64):
                sc = np.zeros((n, L), dtype=np.uint64)
                sc[:, 0] = 1
                sc[1::2, 0] = other
                p = pts[:n].copy()
                if other == 1:
                    p[1] = p[0]                       # the pseudo bucket holds a point twice
                gpu_ctx.msm_config(c, 0)
                try:
                    check(co, gpu_ctx, cid, grp, p, sc, modes=(-1,))
                finally:
                    gpu_ctx.msm_config(0, 0)


@pytest.mark.parametrize("cid,grp", [(0, 1), (1, 2), (2, 1), (2, 2), (3, 2)])
def test_edge_cases(co, gpu_ctx, cid, grp):
    """ragged / degenerate inputs: duplicate bases (doubling branch), flagged infinities, scalars 0, 1, r-1,
    2^c - 1, 2^c around every plausible window size, one giant bucket, sub-ranges of a resident query."""
    fr = co.CURVE_FR[cid]
    n = 400
    pts = co.gen_points(cid, grp, n, seed=2)
    pts[10:20] = pts[9]                      # ten copies of one point
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 50, 399]] = 1
    sc = co.gen_scalars(fr, n, seed=3)
    L = sc.shape[1]
    rm1 = co.fp_op(fr, "to_canonical", co.fp_op(fr, "neg", co.fp_op(fr, "from_canonical", np.array([[1] + [0] * (L - 1)], dtype=np.uint64))))[0]
    sc[1] = rm1
    sc[2] = 0
    sc[3] = 0; sc[3, 0] = 1
    for k, c in enumerate((6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16)):
        sc[30 + 2 * k] = 0; sc[30 + 2 * k, 0] = (1 << c) - 1
        sc[31 + 2 * k] = 0; sc[31 + 2 * k, 0] = 1 << c
    sc[9:20] = sc[9]                         # same point, same scalar: accumulator equals the incoming base
    sc[100:300] = 0; sc[100:300, 0] = 5      # 200 entries in one bucket
    check(co, gpu_ctx, cid, grp, pts, sc, inf=inf)
    check(co, gpu_ctx, cid, grp, pts, sc, inf=inf, offset=7, n=150, modes=(-1, 0))
    check(co, gpu_ctx, cid, grp, pts, sc, inf=inf, offset=399, n=1, modes=(-1,))
    sc[:] = 0
    check(co, gpu_ctx, cid, grp, pts, sc, inf=inf, modes=(-1, 0))


@pytest.mark.parametrize("cid,grp,n", [(0, 1, 20000), (1, 2, 700)])
def test_sort_strategies_and_slot_overflow(co, gpu_ctx, cid, grp, n):
    """the LDS partition sort (default), the single-pass binning and its on-device fallback when a bucket overflows
    its slots (many equal scalars), and the two-pass counting sort must all give the oracle's value."""
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=61)
    for variant in ("uniform", "witness", "heavy"):
        sc = co.gen_scalars(fr, n, seed=62, dist=1 if variant == "witness" else 0)
        if variant == "heavy":
            sc[: n // 2] = sc[0]          # n/2 equal scalars: every window has one bucket far above its slot capacity
        want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=8))
        for sort_mode in (0, 1, 2):
            gpu_ctx.msm_set_sort(sort_mode)
            for pre in (-1, 0):
                gpu_ctx.set_precompute(pre)
                b = gpu_ctx.bases_upload(cid, grp, pts)
                got = co.to_affine(cid, grp, gpu_ctx.msm(b, sc))
                b.free()
                assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), (variant, sort_mode, pre)
    gpu_ctx.msm_set_sort(0)
    gpu_ctx.set_precompute(-1)


@pytest.mark.parametrize("cid", [2, 3])
def test_pair_tree_accumulation(co, gpu_ctx, cid):
    """pcdhip_msm_set_accumulate(2): the 753-bit G1 accumulation as a pair tree of affine additions with shared inversions
    (msm.hip.h msm_pair_tree_kernel, Fp::inv_gcd) -- forced at small sizes with small chunks so that runs cross chunk edges, levels
    stop at different depths and every special pair occurs: equal points (doubling), opposite points (the run cancels), flagged
    infinities, one giant bucket, all scalars equal to one.  Bit-exact against the oracle and against the running-sum form."""
    grp, fr = 1, co.CURVE_FR[cid]
    n = 1500
    pts = co.gen_points(cid, grp, n, seed=90 + cid)
    w = pts.shape[1] // 2
    pts[10:20] = pts[9]                       # ten copies of one point ...
    pts[21] = pts[20]
    pts[21, w:] = co.fp_op(co.CURVE_FQ[cid], "neg", pts[20:21, w:])[0]   # ... and P next to -P
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 50, 51, 1499]] = 1
    cases = {}
    sc = co.gen_scalars(fr, n, seed=7)
    sc[9:22] = sc[9]                          # same scalar on the copies and on P, -P: equal / opposite points meet inside a bucket
    sc[100:700] = 0; sc[100:700, 0] = 5       # 600 entries in one bucket: a run over several chunks
    sc[50:52] = sc[49]                        # infinities next to a finite point of the same bucket
    cases["mixed"] = sc
    cases["witness"] = co.gen_scalars(fr, n, seed=8, dist=1)
    one = np.zeros_like(sc); one[:, 0] = 1
    cases["ones"] = one
    same = np.zeros_like(sc); same[:] = sc[3]
    cases["one scalar"] = same
    try:
        for name, s in cases.items():
            want = co.to_affine(cid, grp, co.msm(cid, grp, pts, s, inf=inf, nthreads=8))
            for pre in (-1, 0):
                gpu_ctx.set_precompute(pre)
                b = gpu_ctx.bases_upload(cid, grp, pts, inf)
                for mode, chunk, min_pairs in ((1, 0, 0), (2, 64, 2), (2, 33, 1), (2, 200, 12), (2, 0, 0)):
                    gpu_ctx.msm_set_accumulate(mode, chunk, min_pairs)
                    got = co.to_affine(cid, grp, gpu_ctx.msm(b, s))
                    assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), (name, pre, mode, chunk, min_pairs)
                b.free()
    finally:
        gpu_ctx.msm_set_accumulate(0)
        gpu_ctx.set_precompute(-1)


def test_points_sum_and_device_scalars(co, gpu_ctx):
    cid, grp, n = 0, 1, 3000
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=21)
    sc = co.gen_scalars(fr, n, seed=22)
    b = gpu_ctx.bases_upload(cid, grp, pts)
    sb = gpu_ctx.buf_upload(fr, sc)
    full = gpu_ctx.msm(b, sb)
    # shard + combine (the multi-GPU step) on one device
    h = n // 3
    parts = np.stack([gpu_ctx.msm(b, sc[:h], offset=0, n=h), gpu_ctx.msm(b, sc[h:], offset=h, n=n - h)])
    comb = gpu_ctx.points_sum(cid, grp, parts)
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=8))
    assert np.array_equal(co.to_affine(cid, grp, full)[0], want[0])
    assert np.array_equal(co.to_affine(cid, grp, comb)[0], want[0])
    assert np.array_equal(gpu_ctx.to_affine(cid, grp, full)[0], want[0])
    b.free(); sb.free()


@pytest.mark.parametrize("cid,grp,world", [(0, 1, 8), (1, 2, 5), (2, 1, 3), (0, 2, 70)])
def test_device_resident_exchange(co, gpu_ctx, cid, grp, world):
    """The N > 1 exchange without host round trips (pcdhip_msm_dev_partial / pcdhip_points_sum_dev): partials written
    into device memory that stands in for the RCCL buffers, summed there; also infinity partials and n > 64 points."""
    import torch
    fr = co.CURVE_FR[cid]
    n = 64 * world
    pts = co.gen_points(cid, grp, n, seed=31)
    sc = co.gen_scalars(fr, n, seed=32)
    sc[:64] = 0                                               # the first shard sums to infinity
    limbs = 3 * co.point_words(cid, grp) // 2
    recv = torch.zeros(world * limbs, dtype=torch.int64, device="cuda:0")
    b = gpu_ctx.bases_upload(cid, grp, pts)
    for r in range(world):
        sb = gpu_ctx.buf_upload(fr, sc[64 * r:64 * (r + 1)])
        gpu_ctx.msm_partial_to_device(b, sb, recv.data_ptr() + 8 * limbs * r, offset=64 * r, n=64)
        gpu_ctx.sync()
        sb.free()
    got = gpu_ctx.points_sum_device(cid, grp, recv.data_ptr(), world)
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=8))
    assert np.array_equal(co.to_affine(cid, grp, got)[0], want[0])
    parts = recv.cpu().numpy().view(np.uint64).reshape(world, limbs)
    assert co.to_affine(cid, grp, parts[:1])[1][0] == 1       # shard 0: the point at infinity
    b.free()


@pytest.mark.parametrize("dist", [0, 1])
def test_full_size_2_20(co, gpu_ctx, dist):
    """BASELINE size (MNT4-298 G1, n = 2^20): direct parity with the multi-threaded oracle, plus linearity
    MSM(k) + MSM(k') = MSM(k + k')."""
    cid, grp, n = 0, 1, 1 << 20
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=0x5043443031)
    sc = co.gen_scalars(fr, n, seed=0x5043443032, dist=dist)
    b = gpu_ctx.bases_upload(cid, grp, pts)
    got = gpu_ctx.msm(b, sc)
    want = co.msm(cid, grp, pts, sc, nthreads=32)
    assert np.array_equal(co.to_affine(cid, grp, got)[0], co.to_affine(cid, grp, want)[0])
    sc2 = co.gen_scalars(fr, n, seed=77, dist=0)
    ssum = co.fp_op(fr, "to_canonical", co.fp_op(fr, "add", co.fp_op(fr, "from_canonical", sc), co.fp_op(fr, "from_canonical", sc2)))
    lhs = co.jac_add(cid, grp, got, gpu_ctx.msm(b, sc2))
    assert np.array_equal(co.to_affine(cid, grp, lhs)[0], co.to_affine(cid, grp, gpu_ctx.msm(b, ssum))[0])
    b.free()


def test_c_abi_error_behaviour(co, gpu_ctx):
    """Every entry point returns a negative PCDHIP_E_* code on bad input (nothing aborts or throws across the ABI)."""
    import ctypes as C
    from pcd_amd import capi
    lib = capi.lib()
    ctx = gpu_ctx._ctx
    h = C.c_void_p()
    one = np.zeros((1, 10), dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.pcdhip_bases_upload(ctx, 9, 1, P(one), None, C.c_size_t(1), C.byref(h)) == -1          # bad curve id
    assert lib.pcdhip_bases_upload(ctx, 0, 3, P(one), None, C.c_size_t(1), C.byref(h)) == -1          # bad group id
    assert lib.pcdhip_bases_upload(ctx, 0, 1, None, None, C.c_size_t(1), C.byref(h)) == -1            # null points
    assert lib.pcdhip_init(99, C.byref(h)) == -1                                                        # no such device
    pts = co.gen_points(0, 1, 8, seed=1)
    b = gpu_ctx.bases_upload(0, 1, pts)
    sc = co.gen_scalars(1, 8, seed=2)
    out = np.zeros(15, dtype=np.uint64)
    assert lib.pcdhip_msm(ctx, b._h, C.c_size_t(4), P(sc), C.c_size_t(8), P(out)) == -1                # range beyond the handle
    assert lib.pcdhip_msm(ctx, b._h, C.c_size_t(0), None, C.c_size_t(8), P(out)) == -1                  # null scalars
    wrong = gpu_ctx.buf_upload(0, sc)                                                                   # scalars of the wrong field
    assert lib.pcdhip_msm_dev(ctx, b._h, C.c_size_t(0), wrong._h, C.c_size_t(0), C.c_size_t(8), P(out)) == -1
    assert lib.pcdhip_fft(ctx, 0, P(np.zeros((4, 5), dtype=np.uint64)), 31, 0, 0) == -1
    assert lib.pcdhip_fft(ctx, 0, P(np.zeros((4, 5), dtype=np.uint64)), 18, 0, 0) == -2                # above the 2-adicity
    assert lib.pcdhip_fft_general(ctx, 1, P(np.zeros((7 * 8, 5), dtype=np.uint64)), C.c_size_t(56), 0, 0) == -2  # no 7-subgroup support on the main field
    assert lib.pcdhip_set_precompute(ctx, 1) == -1 and lib.pcdhip_msm_config(ctx, 99, 0) == -1
    assert lib.pcdhip_msm_set_accumulate(ctx, 3, 0, 0) == -1 and lib.pcdhip_msm_set_accumulate(ctx, 2, 1, 0) == -1 and lib.pcdhip_msm_set_accumulate(ctx, 2, 2000, 0) == -1
    # n = 0 is legal everywhere: the identity
    xy, inf = gpu_ctx.to_affine(0, 1, gpu_ctx.msm(b, sc[:0], n=0))
    assert inf[0] == 1
    b.free(); wrong.free()


def test_submit_collect_pipelined(co, gpu_ctx):
    """pcdhip_msm_submit / collect: four independent MSMs in flight (different bases, sizes, offsets, groups) return what the
    one-at-a-time calls return; a fifth submission is refused until a ticket is collected; an unreduced scalar surfaces at collect."""
    ctx = gpu_ctx
    fr = co.CURVE_FR[0]
    n = 30000
    pts = co.gen_points(0, 1, 3 * n, seed=1501)
    pts2 = co.gen_points(0, 2, n, seed=1502)
    sc = co.gen_scalars(fr, 3 * n, seed=1503)
    b1 = ctx.bases_upload(0, 1, pts)
    b2 = ctx.bases_upload(0, 2, pts2)
    sb = ctx.buf_upload(fr, sc)
    jobs = [(b1, 0, n), (b1, 0, 3 * n), (b2, 0, n), (b1, n + 7, n), (b2, 5, 100), (b1, 0, 5000), (b1, 2 * n, n)]
    want = [co.to_affine(b.curve, b.group, ctx.msm(b, sb, offset=off, n=cnt)) for b, off, cnt in jobs]
    for rounds in range(2):
        got, pending = [], []
        for b, off, cnt in jobs:
            pending.append(ctx.msm_submit(b, sb, offset=off, n=cnt))
            if len(pending) == 4:
                with pytest.raises(Exception):
                    ctx.msm_submit(b, sb, offset=off, n=cnt)            # four tickets are outstanding
                got.append(ctx.msm_collect(pending.pop(0)))
        while pending:
            got.append(ctx.msm_collect(pending.pop(0)))
        for (b, off, cnt), g, w in zip(jobs, got, want):
            a = co.to_affine(b.curve, b.group, g)
            assert np.array_equal(a[0], w[0]) and np.array_equal(a[1], w[1]), (off, cnt, rounds)
    # a proof may not start while tickets are outstanding (it uses the same side streams and workspaces)
    r = co.synthetic_r1cs(fr, 200, 2, seed=1510)
    keys = co.groth16_setup(0, r, co.gen_field(fr, 5, seed=1511), nthreads=4)
    rs = co.gen_field(fr, 2, seed=1512)
    pk = ctx.g16_pk_upload(keys.host_struct(), 0)
    t = ctx.msm_submit(b1, sb, n=1000)
    with pytest.raises(Exception):
        ctx.groth16_prove(pk, r, rs[0], rs[1])
    ctx.msm_collect(t)
    proof, _ = ctx.groth16_prove(pk, r, rs[0], rs[1])
    assert np.array_equal(proof, co.groth16_prove(keys, r, rs[0], rs[1], nthreads=4)[0])
    pk.free()
    bad = sc[:64].copy(); bad[3, -1] = 1 << 60
    bb = ctx.buf_upload(fr, bad)
    t = ctx.msm_submit(b1, bb, n=64)
    with pytest.raises(Exception):
        ctx.msm_collect(t)
    with pytest.raises(Exception):
        ctx.msm(b1, bb, n=64)                  # the one-at-a-time entry points report it too (PCDHIP_E_ARG)
    with pytest.raises(Exception):
        ctx.msm(b1, bad, n=64)
    for h in (b1, b2, sb, bb):
        h.free()


def test_submit_partial_tickets(co, gpu_ctx):
    """pcdhip_msm_submit_partial / pcdhip_msm_ticket_wait (the pipelined exchange of `bench.py --gpus N`): four shard MSMs in flight,
    each partial left in the slot of the caller's device buffer that its ticket names, a foreign stream ordered behind each by an
    event; the four slots summed on the device == the MSM over the whole range == the oracle.  Twice, so slots are reused."""
    import torch
    from pcd_amd import capi
    ctx = gpu_ctx
    cid, grp, n = 0, 1, 4 * 9000
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=1601)
    sc = co.gen_scalars(fr, n, seed=1602, dist=1)
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=8))
    limbs = 3 * co.point_words(cid, grp) // 2
    b = ctx.bases_upload(cid, grp, pts)
    parts = [ctx.buf_upload(fr, sc[q * 9000:(q + 1) * 9000]) for q in range(4)]
    slots = torch.zeros(4 * limbs, dtype=torch.int64, device="cuda:0")
    ts = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        slots.zero_()
        torch.cuda.synchronize()
        tickets = [ctx.msm_submit_partial(b, parts[q], slots.data_ptr(), 8 * limbs, offset=q * 9000, n=9000) for q in range(4)]
        assert sorted(tickets) == [0, 1, 2, 3]
        with pytest.raises(Exception):
            ctx.msm_submit_partial(b, parts[0], slots.data_ptr(), 8 * limbs, offset=0, n=9000)   # four tickets are outstanding
        for t in tickets:
            ctx.msm_ticket_wait(t, ts)
        gathered = slots.clone()                      # on torch's stream: ordered behind the four MSMs by the events
        ctx.stream_wait(ts, 0)
        got = co.to_affine(cid, grp, ctx.points_sum_device(cid, grp, gathered.data_ptr(), 4))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    with pytest.raises(Exception):
        ctx.msm_ticket_wait(0, ts)                    # not outstanding any more
    # an unreduced scalar on this path is reported by the next submission that reuses the ticket's slot (nobody collects a partial ticket)
    bad = sc[:9000].copy()
    bad[17] = 0xFFFFFFFFFFFFFFFF
    badbuf = ctx.buf_upload(fr, bad)
    t_bad = ctx.msm_submit_partial(b, badbuf, slots.data_ptr(), 8 * limbs, offset=0, n=9000)
    ctx.msm_ticket_wait(t_bad, ts)
    outcomes = []
    for q in range(4):                                # round robin: one of the next four submissions lands on that slot
        try:
            t = ctx.msm_submit_partial(b, parts[q], slots.data_ptr(), 8 * limbs, offset=q * 9000, n=9000)
            outcomes.append(t)
        except capi.PrevTicketError as e:             # a code of its own, and THIS submission is in flight: its ticket is valid
            assert e.ticket == t_bad
            t = e.ticket
            outcomes.append("prev")
        ctx.msm_ticket_wait(t, ts)
    assert outcomes.count("prev") == 1, outcomes
    # ... or polled without submitting (the last step of a loop): the status of a released slot
    t_bad = ctx.msm_submit_partial(b, badbuf, slots.data_ptr(), 8 * limbs, offset=0, n=9000)
    t_ok = ctx.msm_submit_partial(b, parts[1], slots.data_ptr(), 8 * limbs, offset=9000, n=9000)
    with pytest.raises(capi.PcdHipError):
        ctx.msm_ticket_status(t_bad)                  # still outstanding: not a released slot
    ctx.msm_ticket_wait(t_bad, ts)
    ctx.msm_ticket_wait(t_ok, ts)
    with pytest.raises(capi.PrevTicketError):
        ctx.msm_ticket_status(t_bad)
    ctx.msm_ticket_status(t_ok)
    with pytest.raises(capi.PcdHipError):
        ctx.msm_ticket_status(t_ok)                   # the word was consumed
    torch.cuda.synchronize()
    ctx.sync()
    badbuf.free()
    b.free()
    for p in parts:
        p.free()


def test_mad_rate_is_a_plausible_roof(gpu_ctx):
    """pcdhip_mad_rate (the live `roofline_int.peak_live` of bench.py): an MI355X issues a v_mad_u64_u32 per SIMD every ~4.5 cycles -- between
    2e13 and 4.5e13 lane-operations a second over 1024 SIMDs at 2 .. 2.4 GHz; repeated calls agree within 15 %"""
    r = [gpu_ctx.mad_rate() for _ in range(3)]
    assert all(2.0e13 < v < 4.5e13 for v in r), r
    assert max(r) / min(r) < 1.15, r


def test_device_chosen_chunk_and_plan_report(co, gpu_ctx):
    """round 5: the entries per lane of the accumulate kernel are chosen on the device from the REAL list length (msm.hip.h msm_plan_chunk) -- a
    witness-like vector (45 % zeros, 35 % ones) leaves a fifth of the n W entries a uniform one makes, and a short list gets short chunks
    instead of a few lanes with 40 entries each; pcdhip_msm_last_plan reports both.  Results equal the oracle at every size / distribution,
    also with the chunk forced (the host-chosen form of rounds 1-4)."""
    ctx = gpu_ctx
    cid, grp = 0, 1
    fr = co.CURVE_FR[cid]
    for n, dist in ((3000, 0), (70000, 1), (70000, 0), (1 << 18, 1)):
        pts = co.gen_points(cid, grp, n, seed=1700 + n)
        sc = co.gen_scalars(fr, n, seed=1701 + n + dist, dist=dist)
        want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=8))
        b = ctx.bases_upload(cid, grp, pts)
        sb = ctx.buf_upload(fr, sc)
        try:
            ctx.msm_profile(True)
            got = co.to_affine(cid, grp, ctx.msm(b, sb))
            entries, chunk = ctx.msm_last_plan()
            ctx.msm_profile(False)
            c_bits, W, _ = ctx.bases_info(b)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (n, dist)
            assert 16 <= chunk <= 56 and 0 < entries <= n * W, (n, dist, entries, chunk)
            if dist == 1:
                assert entries < 0.3 * n * W, (entries, n * W)      # zeros and ones never enter the list
            else:
                assert entries > 0.9 * n * (W - 1)
            ctx.msm_config(0, 33)                                   # a forced chunk: no device plan
            got = co.to_affine(cid, grp, ctx.msm(b, sb))
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (n, dist, "forced chunk")
        finally:
            ctx.msm_profile(False)
            ctx.msm_config(0, 0)
            b.free(); sb.free()
