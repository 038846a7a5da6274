"""The part of the known-answer bridge (tools/kat_export.py, rust/tests/kat.rs, tools/check_kat.py) that covers what the product only
RECALLS of upstream (VERDICT r03 missing #5): the places where a wrong recalled constant would be self-consistent and wrong.

  x_mixed   `MixedRadixEvaluationDomain::{fft, ifft, coset_fft, coset_ifft}` on 7 * 2^3, 49 * 2^2 (MNT4-298 Fq = MNT6-298 Fr) and 5 * 2^3,
            25 * 2^2 (MNT4-753 Fq = MNT6-753 Fr): small-subgroup base, its root of unity, the coset generator
  x_wm      `LibsnarkReduction::witness_map` over `GeneralEvaluationDomain` for a circuit past the field's 2-adicity (33 000 rows over
            MNT6-753's scalar field, 2-adicity 15 -> the domain 5 * 2^13): which domain upstream PICKS, and h on it
  x_fixed   `FixedBaseMSM::{get_mul_window_size, get_window_table, multi_scalar_mul}` + `batch_normalization_into_affine`, eight groups
  x_consts  per field: multiplicative generator, 2-adic root of unity, small-subgroup base / adicity, large-subgroup root of unity;
            per curve: the G1 / G2 generators -- compared with oracle/params.json (what tools/gen_params*.py derived)
  groth16.* `generate_parameters` with the golden toxic waste (alpha, beta, gamma, delta, tau) and the oracle's generators: every query
            of the key, compared with tests/golden/groth16.npz (already there: this only adds the Rust side that computes them)

  x_setup_inf  (round 5) `generate_parameters` over a system with variables that A / B never mention (coracle.witness_r1cs: a packed word
            appears only in C): WHICH entries of a_query / b_g1_query / b_g2_query upstream leaves as the identity with `infinity = true` --
            what the MSMs' infinity bitmaps (msm.hip.h msm_base_is_inf) and the bench's "consistent" synthetic keys rely on
  x_msm_inf    (round 5) `VariableBaseMSM::multi_scalar_mul` over bases some of which ARE the identity, with non-zero scalars on them (G1 and
            G2 of MNT4-298, G2 of MNT6-753)

Inputs are made from seeds by the C++ oracle at export time and the expected values are recomputed from the same seeds at check time
(the oracle is what the HIP path is pinned to at these sizes by tests/): nothing large is committed."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MIXED = ((0, 7, (56, 196)), (2, 5, (40, 100)))
WM_FIELD, WM_ROWS, WM_CURVE = 2, 33000, 3          # MNT6-753: Fr = field 2 (2-adicity 15)
FIXED_N = 40
SETUP_INF_CURVE, SETUP_INF_ROWS = 1, 300      # MNT6-298, a witness-like system: packed words appear in no row of A or B
MSM_INF_CASES = ((0, 1), (0, 2), (3, 2))
MSM_INF_N = 96


def _co():
    import sys
    sys.path.insert(0, ROOT)
    from oracle import coracle as co
    co.build()
    co.lib()
    return co


def wm_r1cs(co):
    return co.synthetic_r1cs(WM_FIELD, WM_ROWS, 2, seed=7100)


def fixed_case(co, c, g):
    fr = co.CURVE_FR[c]
    base = co.gen_points(c, g, 1, seed=7200 + 10 * c + g)[0]
    sc = co.gen_scalars(fr, FIXED_N, seed=7300 + 10 * c + g)
    sc[0] = 0
    sc[1] = 0
    sc[1, 0] = 1
    p = int(json.load(open(os.path.join(ROOT, "oracle", "params.json")))["fields"][fr]["p"])
    sc[2] = [((p - 1) >> (64 * i)) & ((1 << 64) - 1) for i in range(sc.shape[1])]
    return base, sc


def setup_inf_case(co):
    fr = co.CURVE_FR[SETUP_INF_CURVE]
    r = co.witness_r1cs(fr, SETUP_INF_ROWS, 2, seed=7400)
    toxic = co.gen_field(fr, 5, seed=7401)
    return r, toxic


def msm_inf_case(co, c, g):
    fr = co.CURVE_FR[c]
    pts = co.gen_points(c, g, MSM_INF_N, seed=7500 + 10 * c + g)
    sc = co.gen_scalars(fr, MSM_INF_N, seed=7600 + 10 * c + g)
    inf = (np.arange(MSM_INF_N) % 5 == 2).astype(np.uint8)
    inf[0] = 1
    sc[0] = 0
    sc[0, 0] = 1          # scalar one on an identity base
    return pts, inf, sc


def inputs():
    """[(name, array)]: what the Rust side reads (rust/tests/kat_inputs.txt)"""
    co = _co()
    out = []
    r, toxic = setup_inf_case(co)
    for nm in "abc":
        out += [(f"x_setup_inf.rp_{nm}", getattr(r, "rp_" + nm)), (f"x_setup_inf.col_{nm}", getattr(r, "col_" + nm).astype(np.uint64)),
                (f"x_setup_inf.coeff_{nm}", getattr(r, "coeff_" + nm))]
    out += [("x_setup_inf.z", np.asarray(r.z)), ("x_setup_inf.num_inputs", np.array([r.num_inputs], dtype=np.uint64)), ("x_setup_inf.toxic", toxic)]
    for c, g in MSM_INF_CASES:
        pts, inf, sc = msm_inf_case(co, c, g)
        out += [(f"x_msm_inf.c{c}_g{g}_bases", pts), (f"x_msm_inf.c{c}_g{g}_inf", inf), (f"x_msm_inf.c{c}_g{g}_scalars", sc)]
    for fid, _q, sizes in MIXED:
        for n in sizes:
            out.append((f"x_mixed.f{fid}_n{n}_in", co.gen_field(fid, n, seed=7000 + n)))
    r = wm_r1cs(co)
    for nm in "abc":
        out += [(f"x_wm.rp_{nm}", getattr(r, "rp_" + nm)), (f"x_wm.col_{nm}", getattr(r, "col_" + nm).astype(np.uint64)),
                (f"x_wm.coeff_{nm}", getattr(r, "coeff_" + nm))]
    out += [("x_wm.z", np.asarray(r.z)), ("x_wm.num_inputs", np.array([r.num_inputs], dtype=np.uint64))]
    for c in range(4):
        for g in (1, 2):
            base, sc = fixed_case(co, c, g)
            out += [(f"x_fixed.c{c}_g{g}_base", base), (f"x_fixed.c{c}_g{g}_scalars", sc)]
        out += [(f"x_gens.c{c}_g1", co.generator(c, 1)), (f"x_gens.c{c}_g2", co.generator(c, 2))]
    return out


def expected():
    """{name: array} for every x_* line the Rust side writes"""
    co = _co()
    exp = {}
    r, toxic = setup_inf_case(co)
    keys = co.groth16_setup(SETUP_INF_CURVE, r, toxic, nthreads=4)
    assert keys.a_inf.any() and keys.b_g2_inf.any() and not keys.a_inf.all()
    for nm, fl in (("a_query", "a_inf"), ("b_g1_query", "b_g1_inf"), ("b_g2_query", "b_g2_inf")):
        flags = getattr(keys, fl)
        exp[f"x_setup_inf.{nm}"] = np.where(flags[:, None] != 0, 0, getattr(keys, nm)).astype(np.uint64)   # (kat.rs zeroes the limbs of an identity)
        exp[f"x_setup_inf.{nm}_inf"] = flags
    for c, g in MSM_INF_CASES:
        pts, inf, sc = msm_inf_case(co, c, g)
        xy, rinf = co.to_affine(c, g, co.msm(c, g, pts, sc, inf=inf, nthreads=4))
        exp[f"x_msm_inf.c{c}_g{g}_result_xy"], exp[f"x_msm_inf.c{c}_g{g}_result_inf"] = xy[0] if xy.ndim > 1 else xy, np.atleast_1d(rinf)[:1]
    for fid, q, sizes in MIXED:
        for n in sizes:
            x = co.gen_field(fid, n, seed=7000 + n)
            m = q if n % (q * q) else q * q
            for inv in (0, 1):
                for coset in (0, 1):
                    exp[f"x_mixed.f{fid}_n{n}_i{inv}c{coset}"] = co.fft_general(fid, x, m, inverse=bool(inv), coset=bool(coset), nthreads=4)
    r = wm_r1cs(co)
    exp["x_wm.domain_size"] = np.array([co.domain_size(WM_FIELD, r.num_constraints + r.num_inputs)], dtype=np.uint64)
    exp["x_wm.h"] = co.witness_map(r, nthreads=8)
    for c in range(4):
        for g in (1, 2):
            base, sc = fixed_case(co, c, g)
            xy, inf = co.fixed_base_mul(c, g, base, sc, nthreads=4)
            exp[f"x_fixed.c{c}_g{g}_out_xy"], exp[f"x_fixed.c{c}_g{g}_out_inf"] = xy, inf
        exp[f"x_consts.c{c}_g1_generator"], exp[f"x_consts.c{c}_g2_generator"] = co.generator(c, 1), co.generator(c, 2)
    # field constants: from oracle/params.json (Montgomery limbs via the oracle's own conversion)
    P = json.load(open(os.path.join(ROOT, "oracle", "params.json")))
    for fid, f in enumerate(P["fields"]):
        L = int(f["n64"])
        canon = lambda v: np.array([[(int(v) >> (64 * i)) & ((1 << 64) - 1) for i in range(L)]], dtype=np.uint64)
        exp[f"x_consts.f{fid}_generator"] = co.fp_op(fid, "from_canonical", canon(f["generator"]))[0]
        exp[f"x_consts.f{fid}_two_adic_root"] = co.fp_op(fid, "from_canonical", canon(f["root"]))[0]
        exp[f"x_consts.f{fid}_two_adicity"] = np.array([int(f["two_adicity"])], dtype=np.uint64)
    for fid, q, _ in MIXED:
        exp[f"x_consts.f{fid}_small_subgroup_base"] = np.array([q], dtype=np.uint64)
        exp[f"x_consts.f{fid}_small_subgroup_base_adicity"] = np.array([2], dtype=np.uint64)
    return exp


# x_consts lines whose disagreement does NOT break parity of the hot path, with the reason (check_kat prints it instead of failing)
ADVISORY = {
    "g2_generator": "G2 generators of the 753-bit curves were derived here by cofactor clearing: pcdhip_groth16_setup TAKES the generators as "
                    "arguments (upstream samples them), so a different constant only changes which valid key a default setup makes",
    "g1_generator": "as for g2_generator",
}
