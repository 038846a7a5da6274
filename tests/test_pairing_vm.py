"""CPU: the programs of the wave-per-pairing VM (tools/gen_pairing_vm.py -> pcd_amd/csrc/pairing_vm_gen.h).

  * the generated header is the one the generator writes today (nobody edited either side alone);
  * the programs -- formulas, levelling, register allocation, per-slot banking -- evaluated with Python integers give the textbook
    reduced ate pairing of the test oracle (oracle/pyoracle.py: affine slopes, naive final exponent) on all four curves, for single
    pairings, for products, and with points at both ends of the group;
  * structural bounds the device interpreter relies on (terms per instruction, coefficient weight, register counts).
The device's limb arithmetic for the same programs is checked in tests/test_hostcheck.py::test_vm_pairing (host build) and under
`-m gpu` (tests/test_gpu_pairing_kzg.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def gen():
    import gen_pairing_vm as g
    return g


@pytest.fixture(scope="module")
def envs(gen):
    return [gen.compile_env(gen.build(c)) for c in range(4)]


def test_header_is_current(gen, envs):
    have = open(os.path.join(ROOT, "pcd_amd", "csrc", "pairing_vm_gen.h")).read()
    assert have == gen.emit(envs), "pairing_vm_gen.h is stale: run python tools/gen_pairing_vm.py"


@pytest.mark.parametrize("cid", [0, 1, 2, 3])
def test_programs_compute_the_pairing(gen, envs, cid):
    from oracle import pyoracle as po
    env, cv = envs[cid], po.CURVES[cid]
    F1, a1 = cv.group(1)
    F2, a2 = cv.group(2)
    r = cv.fr.p
    pairs = []
    for s1, s2 in ((12345 + cid, 6789), (1, 1), (r - 1, 2), (777, r - 5)):
        P, Q = po.ec_mul(F1, a1, s1, cv.g1), po.ec_mul(F2, a2, s2, cv.g2)
        pairs.append((P, Q))
    want_prod = cv.Fk.one()
    fs = []
    for P, Q in pairs:
        f, _ = gen.eval_miller(env, (P[0][0], P[1][0]), (list(Q[0]), list(Q[1])))
        gt, _ = gen.eval_final_exp(env, [f])
        want = po.pairing(cv, P, Q)
        assert tuple(gt) == tuple(want)
        want_prod = cv.Fk.mul(want_prod, want)
        fs.append(f)
    gt, _ = gen.eval_final_exp(env, fs)
    assert tuple(gt) == tuple(want_prod)
    # e(P, Q) e(-P, Q) = 1
    P, Q = pairs[0]
    f2, _ = gen.eval_miller(env, (P[0][0], (-P[1][0]) % cv.fq.p), (list(Q[0]), list(Q[1])))
    gt, _ = gen.eval_final_exp(env, [fs[0], f2])
    assert tuple(gt) == tuple(cv.Fk.one())


def test_bounds_the_interpreter_relies_on(gen, envs):
    for env in envs:
        nregs = 2 * env.nslots + len(env.const_values) + len(env.regs) + env.ntemp
        assert env.nslots <= 31 and nregs <= 256   # (operand = 8-bit base register | 6-bit bank bit | 2 table flags; bank bit 31 stays clear)
        stride = (env.N + 3) // 4 * 4
        for setname, names in env.sets.items():            # what one kernel stages in LDS: registers + its own programs + script
            slots = steps = 0
            for n in names:
                for kind, instrs in env.compiled[n]["steps"]:
                    steps += 1
                    slots += sum(2 if (kind == gen.K_LIN and len(t) > 8) else 1 for _, t, _ in instrs)
            assert nregs * stride * 4 + slots * 48 + steps * 12 + len(names) * 12 + len(env.scripts[setname]) <= 64 * 1024, (env.name, setname)
            assert all(isinstance(e, tuple) or e in names for e in env.scripts[setname])
        for name, c in env.compiled.items():
            for kind, instrs in c["steps"]:
                assert 1 <= sum(2 if (kind == gen.K_LIN and len(t) > 8) else 1 for _, t, _ in instrs) <= 64
                for dst, terms, k in instrs:
                    assert k == kind
                    if kind == gen.K_LIN:
                        assert 1 <= len(terms) <= gen.LIN_TERMS and sum(abs(cf) for cf, _ in terms) <= gen.LIN_WEIGHT
                    else:
                        assert len(terms) == 1 and (kind == gen.K_MUL or terms[0][0] == terms[0][1])
        # a product closes on [0, 2p) without a final subtraction while 4 p^2 / R' + p < 2p; a LIN reduces values below 2^12 p
        assert 4 * env.p < (1 << (28 * env.N)) and gen.TMAX == 1 and 2 * gen.LIN_WEIGHT < (1 << 12)


def test_schedule_depth_and_flat_records(gen, envs):
    """What the verification's latency rests on (profiles/DESIGN_history_r01-r05.md section 5, round 4): a doubling and an addition step are three product levels
    deep -- six steps -- on all four curves, no step holds more than 64 instruction slots, and the flat record list the device walks is the
    script with its programs expanded: same steps in the same order, the bank flip on each program's last step, the table selections and
    the inversion attached to the step they precede."""
    import re
    for env in envs:
        for name in ("dbl", "add"):
            steps = env.compiled[name]["steps"]
            assert len(steps) == 6 and [k == gen.K_LIN for k, _ in steps] == [False, True] * 3, (env.name, name, len(steps))
        for c in env.compiled.values():
            for kind, instrs in c["steps"]:
                assert sum(2 if (kind == gen.K_LIN and len(t) > 8) else 1 for _, t, _ in instrs) <= gen.LANES
    text = gen.emit(envs)
    for env in envs:
        for setname in env.sets:
            T = f"{env.name}_{setname}"
            grab = lambda what: [tuple(int(x.strip().rstrip("u"), 0) for x in m.group(1).split(","))
                                 for m in re.finditer(r"\{([^{}]+)\}", re.search(rf"{T}_{what}\[\d+\]\[\d\] = \{{(.*?)\}};", text, re.S).group(1))]
            progs, steps, flat = grab("progs"), grab("steps"), grab("flat")
            names = [n for n in env.sets[setname]]
            want, pending = [], 0
            for e in env.scripts[setname]:
                if isinstance(e, tuple) and e[0] == "inv":
                    pending |= 4
                elif isinstance(e, tuple):
                    pending = (pending & ~0xF2) | 2 | (e[1] << 4)
                else:
                    first, cnt, mask = progs[names.index(e)]
                    for q in range(cnt):
                        k, o, n = steps[first + q]
                        fl = (pending if q == 0 else 0) | (1 if q == cnt - 1 else 0)
                        want.append((k | (fl << 16), o, n, mask if q == cnt - 1 else 0))
                    pending = 0
            nscript = int(re.search(rf"{T}_flat_script_len = (\d+);", text).group(1))
            assert nscript == len(want) and flat[:nscript] == want, T
