"""Proofs over the assignment a verifier circuit actually produces (round 5): >= 70 % of z is 0 or 1, in runs -- the
in-circuit Groth16 verifier of /root/reference src/ec_cycle_pcd/data_structures.rs:269-304 (one per prior message) and :381-389 is bit
decompositions with their booleanity rows (coracle.witness_r1cs).  In one proof this combines what the MSM-level tests cover one at a
time: zeros that never enter a list, ones in the pseudo bucket (msm_merge_ones_kernel, the big-bucket kernels), sorts shared between the
MSMs of a proof, the infinity bitmaps of a real key's a / b queries -- through both assembly forms, every schedule, sharded and not, keys
as a setup makes them and dense ones, on all four curves; every proof byte-equal to the CPU oracle's.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 64)


def _devices(k):
    from pcd_amd import capi
    n = capi.lib().pcdhip_device_count()
    return [i % n for i in range(k)]


def _assignment_shape(co, r):
    z = co.fp_op(r.field, "to_canonical", np.ascontiguousarray(r.z))
    zero = ~z.any(axis=1)
    one = (z[:, 0] == 1) & ~z[:, 1:].any(axis=1)
    return float(zero.mean()), float(one.mean())


@pytest.mark.parametrize("cid,nc", [(0, 40000), (1, 9000), (2, 5000), (3, 4000)])
def test_prove_witness_like_all_forms(co, gpu_ctx, cid, nc):
    fr = co.CURVE_FR[cid]
    r = co.witness_r1cs(fr, nc, 2, seed=4400 + cid)
    fz, fo = _assignment_shape(co, r)
    assert fz + fo >= 0.70 and fz >= 0.35 and fo >= 0.25
    rs = co.gen_field(fr, 2, seed=4410 + cid)
    for consistent in (True, False):
        keys = co.synthetic_keys(cid, r, seed=4420 + cid, consistent=consistent)
        want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
        pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
        try:
            for sched in (0, 1, 2):
                gpu_ctx.groth16_set_schedule(sched)
                for mode in (1, 2):   # folded / chained assembly
                    gpu_ctx.groth16_set_assembly(mode)
                    got, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
                    assert np.array_equal(got, want) and np.array_equal(inf, winf), (cid, consistent, sched, mode)
        finally:
            gpu_ctx.groth16_set_assembly(0)
            gpu_ctx.groth16_set_schedule(0)
            pk.free()


@pytest.mark.parametrize("cid,nc,parts", [(0, 30000, 2), (3, 3000, 3)])
def test_prove_witness_like_sharded(co, gpu_ctx, cid, nc, parts):
    """the same assignment through a multi-device context (logical shards of one device when only one is visible): entry ranges of the
    a' / b' / l' queries per device, the assembly products folded in -- equal to the oracle, with and without resident matrices"""
    from pcd_amd import capi
    fr = co.CURVE_FR[cid]
    r = co.witness_r1cs(fr, nc, 2, seed=4500 + cid)
    rs = co.gen_field(fr, 2, seed=4510 + cid)
    keys = co.synthetic_keys(cid, r, seed=4520 + cid)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    mctx = capi.Context(devices=_devices(parts))
    try:
        pk = mctx.g16_pk_upload(keys.host_struct(), cid)
        got, inf = mctx.groth16_prove(pk, r, rs[0], rs[1])
        assert np.array_equal(got, want) and np.array_equal(inf, winf)
        mctx.g16_pk_set_r1cs(pk, r)
        for sched in (0, 2):
            mctx.groth16_set_schedule(sched)
            got, inf = mctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
            assert np.array_equal(got, want) and np.array_equal(inf, winf), sched
        pk.free()
    finally:
        mctx.close()


def _main_2p20_case(co):
    cid, fr = 0, co.CURVE_FR[0]
    r = co.witness_r1cs(fr, (1 << 20) - 8, 2, seed=4600)
    keys = co.synthetic_keys(cid, r, seed=4601, mt=True)
    rs = co.gen_field(fr, 2, seed=4602)
    return (r, keys, rs), lambda: tuple(co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS))


AT_SIZE = {"prove_c0_2p20_witness_like": _main_2p20_case}   # (tests/golden/gen_at_size.py; conftest.Expect)


def test_main_proof_mnt4_298_2p20_witness_like(co, gpu_ctx, expect):
    """BASELINE configs[1]/[2]'s main proof shape at 2^20 rows with the witness-like assignment, consistent key, resident matrices,
    chained and folded assembly"""
    cid = 0
    (r, keys, rs), want_fn = _main_2p20_case(co)
    fz, fo = _assignment_shape(co, r)
    assert fz + fo >= 0.70
    want, winf = expect("prove_c0_2p20_witness_like", want_fn)
    gpu_ctx.groth16_set_sparse_window(-1)   # opt in to the key's second layout under the automatic rule (9.2 GB more at this size: fits)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    gpu_ctx.groth16_set_sparse_window(0)
    gpu_ctx.g16_pk_set_r1cs(pk, r)
    try:
        for mode in (2, 1):
            gpu_ctx.groth16_set_assembly(mode)
            got, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
            assert np.array_equal(got, want) and np.array_equal(inf, winf), mode
            assert gpu_ctx.groth16_last_plan()[0] == (mode == 2), mode   # the chained form takes the shorter-window copies, the folded one never
    finally:
        gpu_ctx.groth16_set_assembly(0)
        pk.free()


def test_key_plan_report_and_lane_reserve(co, gpu_ctx):
    """pcdhip_g16_pk_info (the window plan of a resident key's five queries: what bench.py counts a proof's executed multiply-adds with) and
    pcdhip_set_lane_reserve (the CUs the accumulate lane of schedule 2 leaves to the other streams: 0 = no mask, 8 = default, 16): same proof."""
    cid, fr = 0, co.CURVE_FR[0]
    r = co.witness_r1cs(fr, 30000, 2, seed=4700)
    keys = co.synthetic_keys(cid, r, seed=4701)
    rs = co.gen_field(fr, 2, seed=4702)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
    try:
        plan = gpu_ctx.g16_pk_info(pk)
        assert set(plan) == {"a", "b_g1", "b_g2", "l", "h"}
        for name, (c_bits, W) in plan.items():
            assert 6 <= c_bits <= 22 and W == (298 + c_bits) // c_bits, (name, c_bits, W)   # signed digits: ceil((bits + 1) / c) windows
        assert plan["a"] == plan["b_g1"] == plan["b_g2"] == plan["l"]                       # the four queries over the assignment are laid out alike
        gpu_ctx.groth16_set_schedule(2)
        for reserve in (0, 16, 8, -1):
            gpu_ctx.set_lane_reserve(reserve)
            got, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
            assert np.array_equal(got, want) and np.array_equal(inf, winf), reserve
    finally:
        gpu_ctx.groth16_set_schedule(0)
        gpu_ctx.set_lane_reserve(-1)
        pk.free()


@pytest.mark.parametrize("cid,nc,bits", [(0, 40000, 12), (1, 9000, 10), (3, 4000, 9)])
def test_sparse_window_plan(co, gpu_ctx, cid, nc, bits):
    """pcdhip_groth16_set_sparse_window (round 5): a key that also carries its a / b / l queries laid out for a shorter window; the prover counts the
    general scalars of the assignment on the device and takes those copies for a witness-like assignment, the ordinary ones for a dense one --
    the proof equals the oracle's either way, in both assembly forms (the folded one never takes them), and a key uploaded with the plan switched
    off behaves as before."""
    fr = co.CURVE_FR[cid]
    r = co.witness_r1cs(fr, nc, 2, seed=4800 + cid)
    rs = co.gen_field(fr, 2, seed=4810 + cid)
    keys = co.synthetic_keys(cid, r, seed=4820 + cid)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    z = co.fp_op(fr, "to_canonical", np.ascontiguousarray(r.z))
    general = int(((z[:, 0] > 1) | z[:, 1:].any(axis=1)).sum())
    # the same system with a dense assignment: the matrices stay, z is replaced by what satisfies nothing in particular -- a proof is a function of
    # (key, matrices, z) whether or not z satisfies the system, and the oracle computes the same function
    import copy
    rd = copy.copy(r)
    rd.z = co.gen_field(fr, r.z.shape[0], seed=4830 + cid)
    want_d, winf_d = co.groth16_prove(keys, rd, rs[0], rs[1], nthreads=THREADS)
    try:
        gpu_ctx.groth16_set_sparse_window(bits)
        pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
        try:
            for mode, expect_sparse in ((2, True), (1, False)):
                gpu_ctx.groth16_set_assembly(mode)
                got, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
                assert np.array_equal(got, want) and np.array_equal(inf, winf), (cid, mode)
                used, counted = gpu_ctx.groth16_last_plan()
                assert used == expect_sparse and (counted == general if mode == 2 else counted == 0), (cid, mode, used, counted, general)
            gpu_ctx.groth16_set_assembly(2)
            got, inf = gpu_ctx.groth16_prove(pk, rd, rs[0], rs[1])
            assert np.array_equal(got, want_d) and np.array_equal(inf, winf_d)
            used, counted = gpu_ctx.groth16_last_plan()
            assert not used and counted * 8 > rd.z.shape[0]
        finally:
            pk.free()
        gpu_ctx.groth16_set_sparse_window(0)
        pk = gpu_ctx.g16_pk_upload(keys.host_struct(), cid)
        try:
            got, inf = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
            assert np.array_equal(got, want) and np.array_equal(inf, winf)
            assert gpu_ctx.groth16_last_plan() == (False, 0)
        finally:
            pk.free()
    finally:
        gpu_ctx.groth16_set_sparse_window(0)   # (the library's default: the second layout is opt-in)
        gpu_ctx.groth16_set_assembly(0)
