"""CPU: what can be established about rust/ without a Rust toolchain (tools/check_rust_boundary.py): acyclic crate graph including the
ark-ec / ark-poly fork edits, an impl for every trait bound of the reference's plug-in seam (/root/reference src/ec_cycle_pcd/mod.rs:24-33,
248-254), extern "C" declarations and repr(C) structs equal to include/pcdhip.h, hook types equal to the registered functions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_rust_boundary as crb  # noqa: E402


def test_boundary_checks_pass():
    assert crb.run(verbose=False) == []


def test_header_parser_sees_the_abi():
    protos, structs = crb.c_prototypes()
    assert len(protos) >= 70
    assert protos["pcdhip_msm"] == ("i32", ["ptr", "ptr", "usize", "ptr", "usize", "ptr"])
    assert protos["pcdhip_fft"] == ("i32", ["ptr", "i32", "ptr", "u32", "i32", "i32"])
    assert protos["pcdhip_domain_size"] == ("usize", ["i32", "usize"])
    assert protos["pcdhip_destroy"] == ("void", ["ptr"])
    assert structs["pcdhip_csr"] == ["u64", "ptr", "ptr", "ptr"]
    assert len(structs["pcdhip_g16_pk_host"]) == 22
    # every export the Python binding lists is a prototype of the header (so the parser misses nothing the library exports)
    from pcd_amd import capi
    assert not [n for n in capi.EXPORTS if n not in protos]


def test_checker_reports_a_cycle_and_a_signature_slip(monkeypatch):
    # round 3's S2 patch: curve-crate types named inside the ark-ec fork
    g, _ = crb.dependency_graph({"ark-ec": "use ark_mnt4_298::G1Affine;"})
    cyc = crb.find_cycle(g)
    assert cyc and "ark-ec" in cyc and "ark-mnt4-298" in cyc
    # a Rust declaration that drifts from the header is reported
    real = crb.rust_externs

    def drifted():
        fns, structs = real()
        ret, args = fns["pcdhip_msm"]
        fns = dict(fns, pcdhip_msm=(ret, args[:-1]))
        structs = dict(structs, pcdhip_csr=["u64", "ptr", "ptr"])
        return fns, structs

    monkeypatch.setattr(crb, "rust_externs", drifted)
    errs, _, _ = crb.check_ffi()
    assert any("pcdhip_msm" in e for e in errs) and any("pcdhip_csr" in e for e in errs)
    # a missing impl is reported
    monkeypatch.setattr(crb, "BOUNDS", crb.BOUNDS + [("HipGroth16", "UniversalSetupSNARK", "n/a", [])])
    assert any("UniversalSetupSNARK" in e for e in crb.check_bounds())
