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


@pytest.mark.parametrize("cid,grp", [(0, 1), (1, 2), (2, 1)])
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
