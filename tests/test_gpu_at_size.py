"""BASELINE.json configs at their stated sizes, HIP path against the CPU oracle, through the C-ABI (bit-exact).

  configs[1]  MNT4-298 / MNT6-298 Groth16 PCD step, 2^16-constraint predicate  -> composed proves: main domain 2^17, help 2^16
              (the two SNARK::prove calls of /root/reference src/ec_cycle_pcd/mod.rs:171,179)
  configs[2]  MNT4-753 / MNT6-753 step at 2^20: G1 MSM 2^20, G2 (Fq2) MSM 2^18, G2 (Fq3) MSM 2^16, and the composed proves
              (main domain 2^20, help on the mixed-radix domain 5 * 2^14 its circuit size forces)
  configs[3]  Marlin KZG multi-MSM at 2^20 over a resident 6 * 2^20-point powers vector (tests/mnt4_marlin.rs:72-75 reaches
              KZG10::commit = prefix MSM + hiding MSM)

Inputs are seeded; the oracle's results come from tests/golden/at_size.npz (written in the build container by tests/golden/gen_at_size.py from the
case builders below; PCD_RECOMPUTE=1 runs the oracle on the spot instead -- seconds to ~1 min per case on the GPU box's host cores -- and checks the file)."""
import os

import numpy as np
import pytest

from conftest import Expect

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 64)


# ---- case builders: seeded inputs + the oracle's expectation as a thunk.  tests/golden/gen_at_size.py runs the thunks in the build container
# ---- and commits the results (tests/golden/at_size.npz); the tests take them through the `expect` fixture (PCD_RECOMPUTE=1: the oracle again).
def _msm_case(co, cid, grp, log_n, dist=0):
    n = 1 << log_n
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=900 + 10 * cid + grp)
    sc = co.gen_scalars(fr, n, seed=901 + 10 * cid + grp, dist=dist)
    return (pts, sc), lambda: tuple(co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, nthreads=THREADS)))


def _prove_case(co, curve, nc, seed):
    fr = co.CURVE_FR[curve]
    r = co.synthetic_r1cs(fr, nc, 2, seed=seed)
    keys = co.synthetic_keys(curve, r, seed=seed + 1)
    rs = co.gen_field(fr, 2, seed=seed + 2)
    return (r, keys, rs), lambda: tuple(co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS))


KZG_SLICES = lambda n: ((0, n), (0, 6 * n), (5 * n - 3, n))


def _kzg_case(co):
    curve, fr, n = 0, co.CURVE_FR[0], 1 << 20
    powers = co.gen_points(curve, 1, 6 * n, seed=41)
    gamma = co.gen_points(curve, 1, n, seed=42)
    polys = co.gen_scalars(fr, 6 * n, seed=43)
    blind = co.gen_scalars(fr, n, seed=44)

    def want():
        wanth = co.msm(curve, 1, gamma, blind, nthreads=THREADS)
        out = []
        for off, length in KZG_SLICES(n):
            w = co.msm(curve, 1, powers[off:off + length], polys[:length], nthreads=THREADS)
            out += list(co.to_affine(curve, 1, co.jac_add(curve, 1, w, wanth)))
        return tuple(out)
    return (powers, gamma, polys, blind), want


def _skewed_wm_case(co):
    fr = co.CURVE_FR[0]
    r = co.skewed_r1cs(fr, (1 << 20) - 8, 2, seed=2020)
    return (r,), lambda: Expect.digest(co.witness_map(r, nthreads=THREADS))


def _mk(f, *a, **k):
    return lambda co: f(co, *a, **k)


AT_SIZE = {
    "msm_c2g1_2p20": _mk(_msm_case, 2, 1, 20), "msm_c2g2_2p18": _mk(_msm_case, 2, 2, 18), "msm_c3g2_2p16": _mk(_msm_case, 3, 2, 16),
    "msm_c0g2_2p20_witness_like": _mk(_msm_case, 0, 2, 20, dist=1), "msm_c1g2_2p17": _mk(_msm_case, 1, 2, 17),
    "prove_c0_2p17": _mk(_prove_case, 0, (1 << 17) - 8, 1100), "prove_c1_2p16": _mk(_prove_case, 1, (1 << 16) - 8, 1110),
    "prove_c3_5x2p14": _mk(_prove_case, 3, (1 << 15) + 20000, 1130), "prove_c2_2p20": _mk(_prove_case, 2, (1 << 20) - 8, 1120),
    "kzg_2p20": _kzg_case, "wm_skewed_2p20_sha256": _skewed_wm_case,
}


def _msm_at_size(co, ctx, expect, key, cid, grp):
    (pts, sc), want_fn = AT_SIZE[key](co)
    fr = co.CURVE_FR[cid]
    b = ctx.bases_upload(cid, grp, pts)
    sb = ctx.buf_upload(fr, sc)
    got = co.to_affine(cid, grp, ctx.msm(b, sb))
    b.free(); sb.free()
    want = expect(key, want_fn)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), key


def test_msm_g1_mnt4_753_2p20(co, gpu_ctx, expect):
    _msm_at_size(co, gpu_ctx, expect, "msm_c2g1_2p20", 2, 1)


def test_msm_g2_fq2_mnt4_753_2p18(co, gpu_ctx, expect):
    _msm_at_size(co, gpu_ctx, expect, "msm_c2g2_2p18", 2, 2)


def test_msm_g2_fq3_mnt6_753_2p16(co, gpu_ctx, expect):
    _msm_at_size(co, gpu_ctx, expect, "msm_c3g2_2p16", 3, 2)


def test_msm_g2_mnt4_298_2p20_witness_like(co, gpu_ctx, expect):
    # the main proof's G2 MSM at config-1/2 scale with a witness-like scalar mix (zeros, ones: the pseudo bucket)
    _msm_at_size(co, gpu_ctx, expect, "msm_c0g2_2p20_witness_like", 0, 2)


def test_msm_g2_fq3_mnt6_298_2p17(co, gpu_ctx, expect):
    _msm_at_size(co, gpu_ctx, expect, "msm_c1g2_2p17", 1, 2)


def _prove_at_size(co, ctx, expect, key, curve):
    (r, keys, rs), want_fn = AT_SIZE[key](co)
    pk = ctx.g16_pk_upload(keys.host_struct(), curve)
    ctx.g16_pk_set_r1cs(pk, r)
    try:
        want, winf = expect(key, want_fn)
        for mode in (2, 1):  # chained and folded assembly: the same proof
            ctx.groth16_set_assembly(mode)
            proof, inf = ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
            assert np.array_equal(proof, want) and np.array_equal(inf, winf), (key, mode)
    finally:
        ctx.groth16_set_assembly(0)
        pk.free()
    return keys.domain_size


def test_config1_pcd_step_298(co, gpu_ctx, expect):
    """configs[1]: 2^16-constraint predicate -> main proof on the 2^17 domain (MNT4-298), help proof on 2^16 (MNT6-298)."""
    assert _prove_at_size(co, gpu_ctx, expect, "prove_c0_2p17", 0) == 1 << 17
    assert _prove_at_size(co, gpu_ctx, expect, "prove_c1_2p16", 1) == 1 << 16


def test_config2_pcd_step_753(co, gpu_ctx, expect):
    """configs[2]: main proof MNT4-753 on the 2^20 domain; help proof MNT6-753 whose scalar field has 2-adicity 15, so a
    52 768-row circuit lands on the mixed-radix domain 5 * 2^14 (ark-poly GeneralEvaluationDomain)."""
    assert _prove_at_size(co, gpu_ctx, expect, "prove_c3_5x2p14", 3) == 5 << 14
    assert _prove_at_size(co, gpu_ctx, expect, "prove_c2_2p20", 2) == 1 << 20


def test_config3_kzg_commit_2p20(co, gpu_ctx, expect):
    """configs[3]: KZG10::commit over prefixes of ONE resident 6n-point powers vector, n = 2^20: a commitment of length n, one
    of length 6n and one at an interior offset (shifted powers of a degree bound), each with its n-point hiding MSM."""
    ctx = gpu_ctx
    curve, fr, n = 0, co.CURVE_FR[0], 1 << 20
    (powers, gamma, polys, blind), want_fn = AT_SIZE["kzg_2p20"](co)
    P = ctx.bases_upload(curve, 1, powers)
    G = ctx.bases_upload(curve, 1, gamma)
    S = ctx.buf_upload(fr, polys)
    B = ctx.buf_upload(fr, blind)
    try:
        wants = expect("kzg_2p20", want_fn)
        for k, (off, length) in enumerate(KZG_SLICES(n)):
            c = ctx.msm(P, S, offset=off, n=length)
            h = ctx.msm(G, B, offset=0, n=n)
            got = co.to_affine(curve, 1, ctx.points_sum(curve, 1, np.stack([c, h])))
            assert np.array_equal(got[0], wants[2 * k]) and np.array_equal(got[1], wants[2 * k + 1]), (off, length)
    finally:
        P.free(); G.free(); S.free(); B.free()


def test_witness_map_skewed_2p20(co, gpu_ctx, expect):
    """2^20 rows of the skewed verifier-circuit shape (2.7 M entries in A, rows up to 4155 entries, 80 % unit coefficients) over the
    MNT4-298 scalar field: the witness map against the oracle, matrices handed over with the call and resident with a key (the
    latter with its stage times printed)"""
    fr = co.CURVE_FR[0]
    (r,), want_fn = AT_SIZE["wm_skewed_2p20_sha256"](co)
    want = expect("wm_skewed_2p20_sha256", want_fn)      # (sha256 of the 42 MB vector h)
    assert np.array_equal(Expect.digest(gpu_ctx.witness_map(fr, r)), want)
    keys = co.synthetic_keys(0, r, seed=2022, mt=True)          # (kept alive: host_struct() points into its arrays)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), 0)
    gpu_ctx.g16_pk_set_r1cs(pk, r)
    try:
        h, ms = gpu_ctx.witness_map_resident(pk, r)
        h2, ms = gpu_ctx.witness_map_resident(pk, r)
        print(f"skewed witness map 2^20 (MNT4-298 Fr): {ms}")
        assert np.array_equal(Expect.digest(h), want) and np.array_equal(Expect.digest(h2), want)
    finally:
        pk.free()
