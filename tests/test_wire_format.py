"""Wire format (pcdhip_serialize_* / pcdhip_deserialize_*: the ark-serialize CanonicalSerialize images of GroupAffine, Proof and
VerifyingKey) against the golden bytes written by the pure-Python oracle (tests/golden/gen_golden.py wire -> wire.npz).  Host-side
code: runs without a GPU."""
import numpy as np
import pytest

from pcd_amd import capi

CURVES = [0, 1, 2, 3]


@pytest.mark.parametrize("cid", CURVES)
@pytest.mark.parametrize("grp", [1, 2])
@pytest.mark.parametrize("comp", [0, 1])
def test_points_golden_and_round_trip(golden, cid, grp, comp):
    g = golden("wire")
    pre = f"c{cid}_g{grp}_"
    xy, inf, want = g[pre + "xy"], g[pre + "inf"], g[pre + f"ser{comp}"].tobytes()
    assert capi.lib().pcdhip_serialized_size(cid, grp, comp) * len(inf) == len(want)
    got = capi.serialize_points(cid, grp, xy, inf, compressed=bool(comp))
    assert got == want
    back, binf = capi.deserialize_points(cid, grp, want, len(inf), compressed=bool(comp))   # compressed: y rebuilt by a square root
    assert np.array_equal(binf, inf)
    assert np.array_equal(back[inf == 0], xy[inf == 0]) and not back[inf == 1].any()


@pytest.mark.parametrize("cid", CURVES)
@pytest.mark.parametrize("comp", [0, 1])
def test_proof_and_vk_golden(golden, cid, comp):
    g = golden("wire")
    p1, p2 = g[f"c{cid}_g1_xy"], g[f"c{cid}_g2_xy"]
    proof = np.concatenate([p1[0], p2[3], p1[4]])
    want = g[f"c{cid}_proof_ser{comp}"].tobytes()
    assert capi.proof_serialize(cid, proof, compressed=bool(comp)) == want
    back, inf = capi.proof_deserialize(cid, want, compressed=bool(comp))
    assert np.array_equal(back, proof) and not inf.any()
    abc = np.stack([p1[0], p1[1], p1[5]])
    wantk = g[f"c{cid}_vk_ser{comp}"].tobytes()
    assert capi.vk_serialize(cid, p1[3], p2[0], p2[1], p2[4], abc, compressed=bool(comp)) == wantk
    vk = capi.vk_deserialize(cid, wantk, compressed=bool(comp), max_inputs=8)
    assert np.array_equal(vk["alpha_g1"], p1[3]) and np.array_equal(vk["beta_g2"], p2[0]) and np.array_equal(vk["gamma_g2"], p2[1])
    assert np.array_equal(vk["delta_g2"], p2[4]) and np.array_equal(vk["gamma_abc_g1"], abc)


def test_rejects_malformed(golden):
    g = golden("wire")
    cid, grp = 0, 1
    good = bytearray(g["c0_g1_ser0"].tobytes()[:76])           # one uncompressed point
    bad = bytearray(good); bad[40] ^= 1                         # y changed: not on the curve
    with pytest.raises(capi.PcdHipError):
        capi.deserialize_points(cid, grp, bytes(bad), 1, compressed=False)
    bad = bytearray(good[:38]); bad[-1] |= 0xC0                 # both flags
    with pytest.raises(capi.PcdHipError):
        capi.deserialize_points(cid, grp, bytes(bad), 1, compressed=True)
    bad = bytearray(b"\xff" * 37 + b"\x3f")                     # x >= p
    with pytest.raises(capi.PcdHipError):
        capi.deserialize_points(cid, grp, bytes(bad), 1, compressed=True)
    # some x is not the abscissa of a point: about half of all x
    rejected = 0
    for k in range(8):
        x = bytearray(38); x[0] = k + 2
        try:
            capi.deserialize_points(cid, grp, bytes(x), 1, compressed=True)
        except capi.PcdHipError:
            rejected += 1
    assert 0 < rejected < 8
    assert capi.lib().pcdhip_serialized_size(7, 1, 1) == 0


@pytest.mark.parametrize("cid", CURVES)
def test_g2_subgroup_check(golden, cid):
    """a point of the twist that is on the curve but outside the prime-order subgroup (the twists have huge cofactors: a point
    rebuilt from a small abscissa is such a point, all but surely) is refused by the checked reads and accepted by the unchecked one;
    a proof whose B is such a point is refused too"""
    size = capi.lib().pcdhip_serialized_size(cid, 2, 1)
    found = None
    for k in range(2, 40):
        x = bytearray(size); x[0] = k
        try:
            xy, inf = capi.deserialize_points(cid, 2, bytes(x), 1, compressed=True, unchecked=True)
        except capi.PcdHipError:
            continue                                            # not an abscissa
        found = (bytes(x), xy)
        break
    assert found is not None
    with pytest.raises(capi.PcdHipError):
        capi.deserialize_points(cid, 2, found[0], 1, compressed=True)
    g = golden("wire")
    p1 = g[f"c{cid}_g1_xy"]
    good = g[f"c{cid}_proof_ser1"].tobytes()
    s1 = capi.lib().pcdhip_serialized_size(cid, 1, 1)
    bad = good[:s1] + found[0] + good[s1 + size:]
    assert len(bad) == len(good)
    capi.proof_deserialize(cid, good, compressed=True)
    with pytest.raises(capi.PcdHipError):
        capi.proof_deserialize(cid, bad, compressed=True)
