"""GPU parity of the HIP pairing (pcdhip_multi_pairing / pcdhip_groth16_verify: ark-ec PairingEngine and
ark-groth16 verify, reference call site src/ec_cycle_pcd/mod.rs:239) and of the KZG-style prefix MSMs of the
Marlin configuration (`KZG10::commit` = MSM over a prefix of the resident powers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1], ids=["wave-per-pairing", "lane-per-pairing"])
def pmode(request, gpu_ctx):
    """both forms of the pairing kernels behind the same entry points (pcdhip_pairing_set_mode): one wave per pairing (default for
    batches up to 4096 pairs: pairing_vm.hip.h) and one lane per pairing (pairing.hip.h)"""
    gpu_ctx.pairing_set_mode(request.param)
    yield request.param
    gpu_ctx.pairing_set_mode(0)


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_pairing_golden(golden, gpu_ctx, pmode, cid):
    g = golden("pairing")
    assert np.array_equal(gpu_ctx.multi_pairing(cid, g[f"c{cid}_p"], g[f"c{cid}_q"]), g[f"c{cid}_gt"])


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_pairing_vs_oracle_bilinear_and_product(co, gpu_ctx, pmode, cid):
    fr = co.CURVE_FR[cid]
    k = co.gen_scalars(fr, 2, seed=31)
    g1, g2 = co.generator(cid, 1), co.generator(cid, 2)
    a_g1 = co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, k[0]))[0][0]
    b_g2 = co.to_affine(cid, 2, co.scalar_mul(cid, 2, g2, k[1]))[0][0]
    e_ab = gpu_ctx.multi_pairing(cid, a_g1, b_g2)
    assert np.array_equal(e_ab, co.pairing(cid, a_g1, b_g2))
    # e(aP, bQ) = e(abP, Q)
    ab = co.fp_op(fr, "to_canonical", co.fp_op(fr, "mul", co.fp_op(fr, "from_canonical", k[:1]), co.fp_op(fr, "from_canonical", k[1:2])))[0]
    ab_g1 = co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, ab))[0][0]
    assert np.array_equal(gpu_ctx.multi_pairing(cid, ab_g1, g2), e_ab)
    # product of pairings with an inverse pair is one: e(aP, bQ) e(-abP, Q) = 1; infinity pairs contribute one
    neg = co.to_affine(cid, 1, co.scalar_mul(cid, 1, ab_g1, co.fp_op(fr, "to_canonical", co.fp_op(fr, "neg", co.fp_op(fr, "from_canonical", np.array([[1] + [0] * (k.shape[1] - 1)], dtype=np.uint64))))[0]))[0][0]
    one = gpu_ctx.multi_pairing(cid, np.stack([a_g1, neg, a_g1]), np.stack([b_g2, g2, b_g2]), g1_inf=[0, 0, 1])
    L = k.shape[1] if cid < 2 else 12
    assert one.reshape(-1, co.FIELD_N64[co.CURVE_FQ[cid]])[1:].any() == False  # noqa: E712  (c0 = 1, rest 0)
    assert np.array_equal(gpu_ctx.multi_pairing(cid, np.zeros((0, a_g1.size), dtype=np.uint64), np.zeros((0, b_g2.size), dtype=np.uint64)), one)


@pytest.mark.parametrize("cid,nc", [(0, 300), (1, 200)])
def test_groth16_verify_on_gpu(co, gpu_ctx, pmode, cid, nc):
    """tests/mnt4_groth16.rs:87 / :119 with BOTH prove and verify on the HIP path."""
    fr = co.CURVE_FR[cid]
    r = co.synthetic_r1cs(fr, nc, 3, seed=41)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=42), nthreads=16)
    rs = co.gen_field(fr, 2, seed=43)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    proof, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
    pk.free()
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    args = (cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
    assert gpu_ctx.groth16_verify(*args, pub, proof)
    bad = pub.copy()
    bad[0, 0] ^= 1
    assert not gpu_ctx.groth16_verify(*args, bad, proof)
    bad_proof = proof.copy()
    w1 = co.point_words(cid, 1)
    bad_proof[:w1] = keys.alpha_g1  # a valid point, wrong proof
    assert not gpu_ctx.groth16_verify(*args, pub, bad_proof)


@pytest.mark.parametrize("cid,nc", [(0, 300), (1, 200), (2, 40)])
def test_groth16_verify_batch(co, gpu_ctx, pmode, cid, nc):
    """The inputs of a merge node verified in one call (one launch for all Miller loops): per-proof answers equal the
    single verification and the oracle's, including a wrong public input, a wrong proof point and an empty batch."""
    import time
    fr = co.CURVE_FR[cid]
    r = co.synthetic_r1cs(fr, nc, 3, seed=61)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=62), nthreads=16)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    k = 8
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    proofs, pubs = [], []
    for i in range(k):
        rs = co.gen_field(fr, 2, seed=70 + i)
        proof, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])     # k different proofs of the same statement
        assert not inf.any()
        proofs.append(proof); pubs.append(pub.copy())
    pk.free()
    pubs[2][0, 0] ^= 1                                               # wrong public input
    w1 = co.point_words(cid, 1)
    proofs[5] = proofs[5].copy(); proofs[5][:w1] = keys.alpha_g1     # a valid point, wrong proof
    args = (cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
    t0 = time.perf_counter()
    got = gpu_ctx.groth16_verify_batch(*args, np.stack(pubs), np.stack(proofs))
    t_batch = time.perf_counter() - t0
    want = np.array([i not in (2, 5) for i in range(k)])
    assert np.array_equal(got, want)
    t0 = time.perf_counter()
    single = np.array([gpu_ctx.groth16_verify(*args, pubs[i], proofs[i]) for i in range(k)])
    t_single = time.perf_counter() - t0
    assert np.array_equal(single, want)
    print(f"curve {cid}: {k} verifications batched {t_batch * 1e3:.1f} ms, one by one {t_single * 1e3:.1f} ms")
    assert gpu_ctx.groth16_verify_batch(*args, np.stack(pubs)[:0], np.stack(proofs)[:0]).shape == (0,)


@pytest.mark.parametrize("cid,nc", [(0, 300), (1, 200), (3, 40)])
def test_process_vk_prepared_and_rlc_batch(co, gpu_ctx, pmode, cid, nc):
    """SNARK::process_vk + verify_with_processed_vk (three Miller loops, one final exponentiation per proof, e(alpha, beta) cached) and
    the random-linear-combination batch with ONE shared final exponentiation: accept a batch of valid proofs, reject a batch with a
    wrong public input / a tampered proof; the prepared answers equal the plain ones and the oracle's."""
    fr = co.CURVE_FR[cid]
    r = co.synthetic_r1cs(fr, nc, 3, seed=161)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=162), nthreads=16)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    k = 5
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    proofs = []
    for i in range(k):
        rs = co.gen_field(fr, 2, seed=170 + i)
        proofs.append(gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])[0])
        assert co.groth16_verify(keys, np.ascontiguousarray(r.z[1:r.num_inputs]), proofs[-1])
    pk.free()
    pubs = np.stack([pub] * k)
    proofs = np.stack(proofs)
    pvk = gpu_ctx.process_vk(cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1)
    rho = np.random.default_rng(5).integers(1, 1 << 62, size=(k, 2), dtype=np.uint64)
    try:
        assert gpu_ctx.groth16_verify_prepared(pvk, pubs, proofs).all()
        assert gpu_ctx.groth16_verify_batch_rlc(pvk, pubs, proofs, rho)
        bad_pub = pubs.copy(); bad_pub[3, 0, 0] ^= 1
        assert np.array_equal(gpu_ctx.groth16_verify_prepared(pvk, bad_pub, proofs), np.arange(k) != 3)
        assert not gpu_ctx.groth16_verify_batch_rlc(pvk, bad_pub, proofs, rho)
        w1 = co.point_words(cid, 1)
        bad_pr = proofs.copy(); bad_pr[1, :w1] = keys.alpha_g1            # a valid point, wrong proof
        assert np.array_equal(gpu_ctx.groth16_verify_prepared(pvk, pubs, bad_pr), np.arange(k) != 1)
        assert not gpu_ctx.groth16_verify_batch_rlc(pvk, pubs, bad_pr, rho)
        assert gpu_ctx.groth16_verify_batch_rlc(pvk, pubs[:1], proofs[:1], rho[:1])
        # flagged infinities (A of proof 2, C of proof 4; whatever the flagged coordinates hold is ignored): e(O, .) = 1, so those two fail,
        # in the single-trip form of the wave-per-pairing kernels and in the lane-per-pairing form alike
        pinf = np.zeros((k, 3), dtype=np.uint8); pinf[2, 0] = 1; pinf[4, 2] = 1
        assert np.array_equal(gpu_ctx.groth16_verify_prepared(pvk, pubs, proofs, proofs_inf=pinf), ~np.isin(np.arange(k), (2, 4)))
        if cid == 0:
            # a batch past the wave-per-pairing limit (3 x 1366 > 4096 pairings: the lane-per-pairing kernels, affine accumulations) --
            # this size used to return an error in the default configuration
            big = 1400
            idx = np.arange(big) % k
            bp, bq = pubs[idx].copy(), proofs[idx].copy()
            bp[1001, 0, 0] ^= 1
            bq[37, :w1] = keys.alpha_g1
            got = gpu_ctx.groth16_verify_prepared(pvk, bp, bq)
            want = np.ones(big, dtype=bool); want[1001] = False; want[37] = False
            assert np.array_equal(got, want)
        with pytest.raises(Exception):
            gpu_ctx.groth16_verify_batch_rlc(pvk, pubs, proofs, np.zeros((k, 2), dtype=np.uint64))   # zero is not a challenge
    finally:
        pvk.free()


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_pairing_batch_both_forms_agree(co, gpu_ctx, cid):
    """70 pairs in groups (more than one wave's worth of lanes, infinities at both ends, a group per two pairs): the wave-per-pairing
    and the lane-per-pairing kernels return the same GT elements, and single pairs equal the oracle"""
    n = 70 if cid < 2 else 10
    fr = co.CURVE_FR[cid]
    k = co.gen_scalars(fr, 2 * n, seed=211 + cid)
    g1, g2 = co.generator(cid, 1), co.generator(cid, 2)
    ps = np.stack([co.to_affine(cid, 1, co.scalar_mul(cid, 1, g1, k[i]))[0][0] for i in range(n)])
    qs = np.stack([co.to_affine(cid, 2, co.scalar_mul(cid, 2, g2, k[n + i]))[0][0] for i in range(n)])
    inf1 = np.zeros(n, dtype=np.uint8); inf2 = np.zeros(n, dtype=np.uint8)
    inf1[0] = 1; inf2[n - 1] = 1; inf1[5] = inf2[5] = 1
    outs = []
    for mode in (0, 1):
        gpu_ctx.pairing_set_mode(mode)
        try:
            whole = gpu_ctx.multi_pairing(cid, ps, qs, g1_inf=inf1, g2_inf=inf2)
            singles = [gpu_ctx.multi_pairing(cid, ps[i], qs[i]) for i in (1, 2)]
        finally:
            gpu_ctx.pairing_set_mode(0)
        outs.append((whole, singles))
    assert np.array_equal(outs[0][0], outs[1][0])
    for a, b in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, b)
    assert np.array_equal(outs[0][1][0], co.pairing(cid, ps[1], qs[1]))
    assert np.array_equal(outs[0][1][1], co.pairing(cid, ps[2], qs[2]))


def test_proof_wire_round_trip_verifies(co, gpu_ctx):
    """a GPU-made proof and its key leave as CanonicalSerialize bytes (compressed), come back, and still verify"""
    from pcd_amd import capi
    cid, fr = 1, co.CURVE_FR[1]
    r = co.synthetic_r1cs(fr, 150, 2, seed=181)
    keys = co.groth16_setup(cid, r, co.gen_field(fr, 5, seed=182), nthreads=8)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    rs = co.gen_field(fr, 2, seed=183)
    proof, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
    pk.free()
    blob = capi.proof_serialize(cid, proof, inf)
    vkb = capi.vk_serialize(cid, keys.alpha_g1, keys.beta_g2, keys.gamma_g2, keys.delta_g2, keys.gamma_abc_g1, keys.gamma_abc_inf)
    assert len(blob) == 2 * 38 + 3 * 38                        # MNT6-298: G1 38 B, G2 over Fq3 114 B
    proof2, inf2 = capi.proof_deserialize(cid, blob)
    vk = capi.vk_deserialize(cid, vkb, max_inputs=16)
    assert np.array_equal(proof2, proof) and np.array_equal(inf2, inf)
    pub = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z[1:r.num_inputs]))
    assert gpu_ctx.groth16_verify(cid, vk["alpha_g1"], vk["beta_g2"], vk["gamma_g2"], vk["delta_g2"], vk["gamma_abc_g1"], pub, proof2)


def test_kzg_prefix_commitments(co, gpu_ctx):
    """Marlin/KZG10 shape: one resident `powers_of_g`, commitments are MSMs over prefixes of it (coefficients with
    leading zeros skipped by the caller: offset), plus a hiding MSM over `powers_of_gamma_g`."""
    cid, n = 0, 6000
    fr = co.CURVE_FR[cid]
    powers = co.gen_points(cid, 1, n, seed=51)
    gamma_powers = co.gen_points(cid, 1, 1000, seed=52)
    b = gpu_ctx.bases_upload(cid, 1, powers)
    bg = gpu_ctx.bases_upload(cid, 1, gamma_powers)
    for deg, skip in ((999, 0), (4095, 17), (5999, 0), (10, 3)):
        coeffs = co.gen_scalars(fr, deg + 1 - skip, seed=deg)
        rnd = co.gen_scalars(fr, 1000, seed=deg + 1)
        com = gpu_ctx.msm(b, coeffs, offset=skip, n=deg + 1 - skip)
        hide = gpu_ctx.msm(bg, rnd)
        got = gpu_ctx.points_sum(cid, 1, np.stack([com, hide]))
        want = co.jac_add(cid, 1, co.msm(cid, 1, powers[skip:deg + 1], coeffs, nthreads=8), co.msm(cid, 1, gamma_powers, rnd, nthreads=8))
        assert np.array_equal(co.to_affine(cid, 1, got)[0], co.to_affine(cid, 1, want)[0])
    b.free(); bg.free()
