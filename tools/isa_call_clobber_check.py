"""Developer tool (no GPU needed): does a caller keep a value in a register across a call that the callee overwrites?

    python tools/isa_call_clobber_check.py file.s [substring-of-kernel-name ...]

hipcc compiles the non-inlined field products (mailbox calls) with interprocedural register allocation: a caller may keep values in
any register the callee is recorded not to touch.  This tool re-derives, from the final assembly (`hipcc -S --cuda-device-only`), the set of
SGPRs / VGPRs / AGPRs every callee writes without restoring, walks the control-flow graph of every kernel and reports each READ of a
register whose reaching definition lies before a call that clobbers it.  It found the round-3 "one wrong limb" of the Fq3-753 mailbox
addition.  Heuristic operand parsing: a report is a lead to check in the listing, not a proof.
"""
import collections
import re
import subprocess
import sys

REG = re.compile(r"\b([sva])(\d+)\b|\b([sva])\[(\d+):(\d+)\]")


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1)]
    if re.search(r"\bvcc\b", tok):
        out += [("s", 106), ("s", 107)]
    if re.search(r"\bvcc_lo\b", tok):
        out.append(("s", 106))
    if re.search(r"\bvcc_hi\b", tok):
        out.append(("s", 107))
    return out


NO_DST = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setpc", "s_bitcmp", "s_sleep", "s_sethalt",
          "s_setprio", "s_sendmsg", "s_trap", "s_icache", "s_dcache", "s_setreg", "s_store", "s_atomic", "buffer_store", "global_store",
          "scratch_store", "flat_store", "ds_write", "ds_store", "v_cmpx", "s_code_end", "s_set_gpr", "v_nop", "s_setvskip", "global_atomic",
          "buffer_wbl2", "buffer_inv", "s_ttrace", "s_inst_prefetch", "s_clause", "s_delay", "ds_nop", "ds_gws", "s_version")
TWO_DST = ("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale")


def split_ops(l):
    parts = l.split(None, 1)
    op = parts[0]
    args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
    # re-join bracketed register ranges split by nothing (ranges contain ':' not ','), modifiers like offset:16 stay in the last operand
    return op, args


def defs_uses(l):
    op, args = split_ops(l)
    if op.startswith("s_swappc"):
        return [("s", 30), ("s", 31)], regs_of(args[1]) if len(args) > 1 else []
    if op.startswith(NO_DST):
        u = []
        for a in args:
            u += regs_of(a)
        return [], u
    d, u = [], []
    nd = 2 if op.startswith(TWO_DST) and (op.endswith("_e64") or op.startswith(("v_mad_u64", "v_mad_i64", "v_div_scale")) or len(args) >= 4) else 1
    # e32 carry ops write vcc implicitly
    if op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co")) and nd == 1:
        d += [("s", 106), ("s", 107)]
    if op.startswith("v_cmp") and op.endswith("_e32"):
        d += [("s", 106), ("s", 107)]
        for a in args:
            u += regs_of(a)
        return d, u
    for i, a in enumerate(args):
        if i < nd:
            d += regs_of(a)
        else:
            u += regs_of(a)
    if op.startswith(("v_cndmask_b32_e32", "v_addc_co_u32_e32", "v_subb_co_u32_e32", "v_subbrev_co_u32_e32")):
        u += [("s", 106), ("s", 107)]
    if op.startswith(("v_writelane", "v_mac", "v_fmac", "v_accvgpr_write")) and op.startswith(("v_writelane", "v_mac", "v_fmac")):
        u += d  # read-modify-write
    if op.startswith("s_cselect") or op.startswith("s_cmov") or op.startswith("s_addc") or op.startswith("s_subb"):
        pass
    return d, u


def parse(path):
    funcs, cur = collections.OrderedDict(), None
    kernels = set()
    for line in open(path):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L") and not line.startswith("\t"):
            cur = m.group(1)
            funcs[cur] = []
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            kernels.add(m.group(1))
        if cur is None:
            continue
        l = line.strip()
        if not l or l.startswith(";"):
            continue
        if l.startswith(".") and not l.startswith(".LBB"):
            continue
        funcs[cur].append(re.sub(r"\s*;.*", "", l))
    return funcs, kernels


def callee_clobbers(code):
    """registers a function writes and does not restore (prologue saves / epilogue restores are matched pairwise)"""
    written = set()
    saved = {}
    restored = set()
    for l in code:
        if l.startswith(".LBB"):
            continue
        d, _ = defs_uses(l)
        written.update(d)
        m = re.match(r"v_accvgpr_write_b32 a(\d+), v(\d+)", l)
        if m:
            saved.setdefault(("v", int(m.group(2))), ("a", int(m.group(1))))
        m = re.match(r"scratch_store_dword off, v(\d+), s32(?: offset:(\d+))?$", l)
        if m:
            saved.setdefault(("v", int(m.group(1))), ("m", int(m.group(2) or 0)))
    for l in code[-400:]:
        m = re.match(r"v_accvgpr_read_b32 v(\d+), a(\d+)", l)
        if m and saved.get(("v", int(m.group(1)))) == ("a", int(m.group(2))):
            restored.add(("v", int(m.group(1))))
        m = re.match(r"scratch_load_dword v(\d+), off, s32(?: offset:(\d+))?$", l)
        if m and saved.get(("v", int(m.group(1)))) == ("m", int(m.group(2) or 0)):
            restored.add(("v", int(m.group(1))))
    return written - restored


def call_targets(code):
    """index of s_swappc -> callee name (resolved through the s_getpc / s_add_u32 sym@rel32@lo idiom, moves and SGPR spills)"""
    sym_at = {}
    out = {}
    lane_sym = {}
    regsym = {}
    for i, l in enumerate(code):
        m = re.match(r"s_add_u32 s(\d+), s\d+, (\S+)@rel32@lo", l)
        if m:
            regsym[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r"s_mov_b64 s\[(\d+):\d+\], s\[(\d+):\d+\]", l)
        if m and int(m.group(2)) in regsym:
            regsym[int(m.group(1))] = regsym[int(m.group(2))]
            continue
        m = re.match(r"v_writelane_b32 v(\d+), s(\d+), (\d+)", l)
        if m and int(m.group(2)) in regsym:
            lane_sym[(int(m.group(1)), int(m.group(3)))] = regsym[int(m.group(2))]
            continue
        m = re.match(r"v_readlane_b32 s(\d+), v(\d+), (\d+)", l)
        if m:
            k = (int(m.group(2)), int(m.group(3)))
            if k in lane_sym:
                regsym[int(m.group(1))] = lane_sym[k]
            else:
                regsym.pop(int(m.group(1)), None)
            continue
        m = re.match(r"s_swappc_b64 s\[30:31\], s\[(\d+):", l)
        if m:
            out[i] = regsym.get(int(m.group(1)))
            continue
        if not l.startswith(".LBB"):
            d, _ = defs_uses(l)
            for k, r in d:
                if k == "s" and not re.match(r"s_addc_u32", l):
                    regsym.pop(r, None)
    return out


def check_kernel(name, code, clob):
    labels = {l[:-1]: i for i, l in enumerate(code) if l.startswith(".LBB")}
    targets = call_targets(code)
    n = len(code)
    # forward dataflow over instructions: state[r] = index of the call that clobbered r since its last definition (absent = fine)
    instate = [None] * n
    work = [(0, {})]
    reports = {}
    visits = 0
    while work:
        i, st = work.pop()
        while i < n:
            visits += 1
            if visits > 5_000_000:
                return reports
            old = instate[i]
            if old is not None:
                merged = dict(old)
                changed = False
                for r, c in st.items():
                    if r not in merged:
                        merged[r] = c
                        changed = True
                if not changed:
                    break
                st = merged
            instate[i] = dict(st)
            l = code[i]
            if l.startswith(".LBB"):
                i += 1
                continue
            d, u = defs_uses(l)
            if l.startswith("s_swappc"):
                callee = targets.get(i)
                cl = clob.get(callee)
                st = dict(st)
                if cl is None:
                    reports.setdefault(("?", i), f"call at {i} to unresolved target")
                else:
                    for r in cl:
                        st[r] = (i, callee)
                for r in d:
                    st.pop(r, None)
                i += 1
                continue
            for r in u:
                if r in st and r[0] in "sva" and not (r[0] == "s" and r[1] >= 106):
                    reports.setdefault((r, st[r][0]), f"{r[0]}{r[1]} read at {i} `{l}` after call at {st[r][0]} to {short(st[r][1])} clobbered it")
            if d:
                st = dict(st)
                for r in d:
                    st.pop(r, None)
            if l.startswith("s_endpgm") or l.startswith("s_setpc"):
                break
            m = re.match(r"s_branch (\S+)", l)
            if m:
                i = labels[m.group(1)]
                continue
            m = re.match(r"s_cbranch_\w+ (\S+)", l)
            if m:
                work.append((labels[m.group(1)], dict(st)))
            i += 1
    return reports


def short(sym):
    if not sym:
        return str(sym)
    return subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("pcd::", "")[:110]


def main():
    path = sys.argv[1]
    pats = sys.argv[2:]
    funcs, kernels = parse(path)
    clob = {f: callee_clobbers(c) for f, c in funcs.items() if f not in kernels}
    bad = 0
    for k in funcs:
        if k not in kernels or not any("s_swappc" in l for l in funcs[k]):
            continue
        if pats and not any(p in k or p in short(k) for p in pats):
            continue
        rep = check_kernel(k, funcs[k], clob)
        print(f"{short(k)}: {len(rep)} suspicious reads")
        for key in sorted(rep, key=lambda t: str(t))[:12]:
            print("   ", rep[key])
        bad += len(rep)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
