"""Multi-device paths on whatever GPUs are visible (BASELINE configs[4]).  With one GPU the device list names it twice:
two logical shards on one device run exactly the code a two-GPU context runs (per-device sub-contexts, peer copies, gather, sum).

  * sharded MSM and sharded Groth16 prove through a multi-device context == the single-device result == the oracle, bit for bit
  * DAG branches: N host threads x N contexts proving different statements concurrently (pcd_amd/dag.py), each == the oracle"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = min(os.cpu_count() or 1, 64)


def _devices(k):
    from pcd_amd import capi
    n = capi.lib().pcdhip_device_count()
    return [i % n for i in range(k)]


@pytest.mark.parametrize("cid,grp,n,parts", [(0, 1, 50000, 2), (1, 2, 3000, 3), (2, 1, 1500, 2), (0, 1, 5, 4)])
def test_sharded_msm(co, gpu_ctx, cid, grp, n, parts):
    from pcd_amd import capi
    fr = co.CURVE_FR[cid]
    pts = co.gen_points(cid, grp, n, seed=700 + n)
    sc = co.gen_scalars(fr, n, seed=701 + n, dist=1)
    inf = np.zeros(n, dtype=np.uint8)
    inf[n // 2] = 1
    want = co.to_affine(cid, grp, co.msm(cid, grp, pts, sc, inf=inf, nthreads=THREADS))
    mctx = capi.Context(devices=_devices(parts))
    try:
        assert capi.lib().pcdhip_ctx_devices(mctx._ctx) == parts
        b = mctx.bases_upload(cid, grp, pts, inf)
        got = co.to_affine(cid, grp, mctx.msm(b, sc))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        # a sub-range that starts inside one shard and ends inside another, and one that misses a shard entirely
        for off, cnt in ((n // 3, n // 2), (0, max(1, n // (2 * parts)))):
            got = co.to_affine(cid, grp, mctx.msm(b, sc[:cnt], offset=off, n=cnt))
            w = co.to_affine(cid, grp, co.msm(cid, grp, pts[off:off + cnt], sc[:cnt], inf=inf[off:off + cnt], nthreads=THREADS))
            assert np.array_equal(got[0], w[0]) and np.array_equal(got[1], w[1]), (off, cnt)
        b.free()
    finally:
        mctx.close()


@pytest.mark.parametrize("curve,nc,parts", [(0, 3000, 2), (1, 900, 3), (3, 40000, 2)])
def test_sharded_groth16_prove(co, gpu_ctx, curve, nc, parts):
    """the merge node's proof on all devices == the single-device proof == the oracle's (curve 3 at 40000 rows: mixed-radix domain)"""
    from pcd_amd import capi
    fr = co.CURVE_FR[curve]
    r = co.synthetic_r1cs(fr, nc, 2, seed=800 + nc)
    keys = co.synthetic_keys(curve, r, seed=801 + nc)
    rs = co.gen_field(fr, 2, seed=802)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    pk = gpu_ctx.g16_pk_upload(keys.host_struct(), curve)
    single, _ = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
    gpu_ctx.groth16_set_schedule(1)               # the witness map first, then all five MSMs: same proof
    single1, _ = gpu_ctx.groth16_prove(pk, r, rs[0], rs[1])
    gpu_ctx.groth16_set_schedule(0)
    pk.free()
    assert np.array_equal(single, want) and np.array_equal(single1, want)
    mctx = capi.Context(devices=_devices(parts))
    try:
        mpk = mctx.g16_pk_upload(keys.host_struct(), curve)
        got, inf = mctx.groth16_prove(mpk, r, rs[0], rs[1])                      # matrices handed over with the call
        assert np.array_equal(got, want) and np.array_equal(inf, winf)
        mctx.g16_pk_set_r1cs(mpk, r)
        got, inf = mctx.groth16_prove(mpk, r, rs[0], rs[1], resident_r1cs=True)  # matrices resident on device 0
        assert np.array_equal(got, want) and np.array_equal(inf, winf)
        mpk.free()
    finally:
        mctx.close()


def test_dag_branches_threads_x_contexts(co, gpu_ctx):
    """four independent branches, each in its own host thread with its own context, concurrently"""
    from pcd_amd import dag
    jobs, wants = [], []
    for i, (curve, nc) in enumerate(((0, 5000), (1, 2000), (0, 700), (1, 4000))):
        fr = co.CURVE_FR[curve]
        r = co.synthetic_r1cs(fr, nc, 2, seed=900 + i)
        keys = co.synthetic_keys(curve, r, seed=910 + i)
        rs = co.gen_field(fr, 2, seed=920 + i)
        wants.append(co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)[0])

        def branch(ctx, curve=curve, r=r, keys=keys, rs=rs):
            pk = ctx.g16_pk_upload(keys.host_struct(), curve)
            out = [ctx.groth16_prove(pk, r, rs[0], rs[1])[0] for _ in range(3)]
            pk.free()
            return out
        jobs.append(branch)
    results = dag.run_branches(jobs, _devices(4))
    for outs, want in zip(results, wants):
        for got in outs:
            assert np.array_equal(got, want)


@pytest.mark.parametrize("curve,nc,parts", [(0, 5000, 3), (1, 900, 3), (3, 40000, 4), (2, 3000, 8)])
def test_witness_map_chains_on_three_devices(co, gpu_ctx, curve, nc, parts):
    """with >= 3 devices and resident matrices the witness map's a / b / c chains run on devices 0 / 1 / 2 and the two vectors
    travel back device to device; the proof equals the all-on-device-0 form and the oracle's (radix-2 and mixed-radix domains, repeated
    proves with a changed assignment -- the buffers of one proof must not leak into the next)"""
    from pcd_amd import capi
    fr = co.CURVE_FR[curve]
    r = co.skewed_r1cs(fr, nc, 2, seed=850 + nc)
    keys = co.synthetic_keys(curve, r, seed=851 + nc)
    rs = co.gen_field(fr, 2, seed=852)
    want, winf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    mctx = capi.Context(devices=_devices(parts))
    try:
        mpk = mctx.g16_pk_upload(keys.host_struct(), curve)
        mctx.g16_pk_set_r1cs(mpk, r)
        for on, sched in ((1, 0), (0, 0), (1, 1), (0, 1), (1, 0)):
            mctx.groth16_set_witness_split(on)
            mctx.groth16_set_schedule(sched)      # 1: every device's MSMs behind its chain of the witness map
            got, inf = mctx.groth16_prove(mpk, r, rs[0], rs[1], resident_r1cs=True)
            assert np.array_equal(got, want) and np.array_equal(inf, winf), (on, sched)
        # another statement under the same key and matrices: scale the witness-independent part by proving with other blinding factors
        rs2 = co.gen_field(fr, 2, seed=853)
        want2, _ = co.groth16_prove(keys, r, rs2[0], rs2[1], nthreads=THREADS)
        got2, _ = mctx.groth16_prove(mpk, r, rs2[0], rs2[1], resident_r1cs=True)
        assert np.array_equal(got2, want2)
        mpk.free()
    finally:
        mctx.close()


def test_bit_identity_at_1_2_4_8_shards(co, gpu_ctx):
    """the same MSM and the same proof through contexts of 1, 2, 4 and 8 shards -- the devices that are visible, round-robin
    (`_devices`: with one GPU every shard is a logical shard of it; with >= 2 the sub-contexts sit on DISTINCT devices and the partial
    results travel by real peer copies) -- must give the same bytes as the single-device context and the oracle.  The test prints which
    case it ran so that the first run on a multi-GPU node says what it covered."""
    from pcd_amd import capi
    ngpu = capi.lib().pcdhip_device_count()
    cid, fr = 0, co.CURVE_FR[0]
    n = 70000
    pts = co.gen_points(cid, 1, n, seed=990)
    sc = co.gen_scalars(fr, n, seed=991, dist=1)
    want_msm = co.to_affine(cid, 1, co.msm(cid, 1, pts, sc, nthreads=THREADS))
    r = co.witness_r1cs(fr, 20000, 2, seed=992)
    keys = co.synthetic_keys(cid, r, seed=993)
    rs = co.gen_field(fr, 2, seed=994)
    want_proof, want_inf = co.groth16_prove(keys, r, rs[0], rs[1], nthreads=THREADS)
    for parts in (1, 2, 4, 8):
        devs = _devices(parts)
        mctx = capi.Context(devices=devs) if parts > 1 else capi.Context(devs[0])
        try:
            b = mctx.bases_upload(cid, 1, pts)
            got = co.to_affine(cid, 1, mctx.msm(b, sc))
            assert np.array_equal(got[0], want_msm[0]) and np.array_equal(got[1], want_msm[1]), parts
            b.free()
            pk = mctx.g16_pk_upload(keys.host_struct(), cid)
            mctx.g16_pk_set_r1cs(pk, r)
            for _ in range(2):
                proof, inf = mctx.groth16_prove(pk, r, rs[0], rs[1], resident_r1cs=True)
                assert np.array_equal(proof, want_proof) and np.array_equal(inf, want_inf), parts
            pk.free()
        finally:
            mctx.close()
    print(f"shard identity 1/2/4/8: {ngpu} device(s) visible -> {'DISTINCT devices, real peer copies' if ngpu >= 2 else 'logical shards of one device'}")
