"""Static checks of the Rust boundary (rust/) that need no Rust toolchain -- there is none in the build container, so the crate has
never met a compiler; these are the properties that can be established by reading:

 1. the crate dependency graph of rust/Cargo.toml plus the edits of the ark-ec / ark-poly forks (rust/s2_patch/<crate>/...) is ACYCLIC
    (round 3's S2 patch named curve-crate types from inside ark-ec: a cycle Cargo rejects; `--self-test` replays that mistake);
 2. every trait bound the reference puts on the types a user plugs into `ECCyclePCDConfig` (/root/reference src/ec_cycle_pcd/mod.rs:24-33)
    and on `CircuitSpecificSetupPCD` (mod.rs:248-254) has an `impl` in rust/src, with the methods / associated types the trait requires;
 3. every `extern "C"` declaration in rust/src matches include/pcdhip.h in name, arity, pointer / integer kind of every parameter and
    of the result, and every `#[repr(C)]` struct matches its C typedef field by field;
 4. the type-erased hook types of the forks (MsmHook / FftHook) match the functions rust/src/s2.rs registers.

    python tools/check_rust_boundary.py            # exit code 0 = all checks pass
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = os.path.join(ROOT, "rust")

# [UPSTREAM] dependency edges of the arkworks 0.3-era crates (their Cargo.toml [dependencies], ark-* only; dev-dependencies do not count
# for cycles between library crates)
UPSTREAM = {
    "ark-std": [],
    "ark-serialize": ["ark-std"],
    "ark-ff": ["ark-std", "ark-serialize"],
    "ark-ec": ["ark-ff", "ark-std", "ark-serialize"],
    "ark-poly": ["ark-ff", "ark-std", "ark-serialize"],
    "ark-relations": ["ark-ff", "ark-std"],
    "ark-snark": ["ark-ff", "ark-std", "ark-relations"],
    "ark-nonnative-field": ["ark-ff", "ark-ec", "ark-std", "ark-relations", "ark-r1cs-std"],
    "ark-r1cs-std": ["ark-ff", "ark-ec", "ark-std", "ark-relations"],
    "ark-mnt4-298": ["ark-ff", "ark-ec", "ark-std", "ark-r1cs-std"],
    "ark-mnt6-298": ["ark-ff", "ark-ec", "ark-std", "ark-r1cs-std", "ark-mnt4-298"],
    "ark-mnt4-753": ["ark-ff", "ark-ec", "ark-std", "ark-r1cs-std"],
    "ark-mnt6-753": ["ark-ff", "ark-ec", "ark-std", "ark-r1cs-std", "ark-mnt4-753"],
    "ark-crypto-primitives": ["ark-ff", "ark-ec", "ark-std", "ark-relations", "ark-snark", "ark-r1cs-std", "ark-nonnative-field"],
    "ark-groth16": ["ark-ff", "ark-ec", "ark-poly", "ark-serialize", "ark-std", "ark-relations", "ark-crypto-primitives", "ark-r1cs-std"],
}


def crate_refs(text):
    """crates a Rust source names: `ark_xxx::`, `use ark_xxx`, `extern crate ark_xxx` (comments stripped)"""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return {m.replace("_", "-") for m in re.findall(r"\b(ark_[a-z0-9_]+)\b", text)}


def dependency_graph(extra_fork_sources=None):
    g = {k: set(v) for k, v in UPSTREAM.items()}
    cargo = open(os.path.join(RUST, "Cargo.toml")).read()
    deps_sec = cargo.split("[dependencies]", 1)[1].split("\n[", 1)[0]
    deps = re.findall(r"^([a-z0-9-]+)\s*=", deps_sec, flags=re.M)
    g["ark-pcd-hip"] = set(deps)
    src = crate_refs("".join(open(os.path.join(RUST, "src", f)).read() for f in os.listdir(os.path.join(RUST, "src")) if f.endswith(".rs")))
    undeclared = {c for c in src if c not in g["ark-pcd-hip"] and c != "ark-pcd-hip"}
    forks = os.path.join(RUST, "s2_patch")
    for crate in sorted(os.listdir(forks)):
        d = os.path.join(forks, crate)
        if not os.path.isdir(d):
            continue
        for dp, _, fs in os.walk(d):
            for f in fs:
                if f.endswith(".rs"):
                    g.setdefault(crate, set()).update(c for c in crate_refs(open(os.path.join(dp, f)).read()) if c != crate)
    for crate, text in (extra_fork_sources or {}).items():
        g.setdefault(crate, set()).update(c for c in crate_refs(text) if c != crate)
    return g, undeclared


def find_cycle(g):
    color, stack = {}, []

    def dfs(u):
        color[u] = 1
        stack.append(u)
        for v in sorted(g.get(u, ())):
            if color.get(v) == 1:
                return stack[stack.index(v):] + [v]
            if v not in color:
                c = dfs(v)
                if c:
                    return c
        stack.pop()
        color[u] = 2
        return None

    for n in sorted(g):
        if n not in color:
            c = dfs(n)
            if c:
                return c
    return None


# (type, trait, reference site, required items of the impl)
BOUNDS = [
    ("HipGroth16", "SNARK", "src/ec_cycle_pcd/mod.rs:28-29 `type MainSNARK: SNARK<MainField>; type HelpSNARK: SNARK<HelpField>`",
     ["type ProvingKey", "type VerifyingKey", "type Proof", "type ProcessedVerifyingKey", "type Error", "fn circuit_specific_setup", "fn prove",
      "fn process_vk", "fn verify_with_processed_vk"]),
    ("HipGroth16VerifierGadget", "SNARKGadget", "src/ec_cycle_pcd/mod.rs:31-32 `type MainSNARKGadget: SNARKGadget<MainField, HelpField, Self::MainSNARK>`",
     ["type ProcessedVerifyingKeyVar", "type VerifyingKeyVar", "type InputVar", "type ProofVar", "type VerifierSize", "fn verifier_size",
      "fn verify_with_processed_vk", "fn verify"]),
    ("HipGroth16", "CircuitSpecificSetupSNARK", "src/ec_cycle_pcd/mod.rs:248-254 `IC::MainSNARK: CircuitSpecificSetupSNARK<MainField>, IC::HelpSNARK: ..`", []),
]
REFERENCE_LINES = {"SNARK": ("src/ec_cycle_pcd/mod.rs", 24, 33), "SNARKGadget": ("src/ec_cycle_pcd/mod.rs", 24, 33),
                   "CircuitSpecificSetupSNARK": ("src/ec_cycle_pcd/mod.rs", 246, 254)}


def impl_block(src, trait, typ):
    m = re.search(r"impl\s*<[^{]*?>\s*" + trait + r"\s*<[^{]*?>\s*for\s+" + typ + r"\s*<[^{]*?>\s*\{", src, flags=re.S)
    if not m:
        return None
    i, depth = m.end(), 1
    while depth and i < len(src):
        depth += {"{": 1, "}": -1}.get(src[i], 0)
        i += 1
    return src[m.end():i - 1]


def check_bounds():
    src = open(os.path.join(RUST, "src", "lib.rs")).read()
    errs = []
    for typ, trait, site, items in BOUNDS:
        body = impl_block(src, trait, typ)
        if body is None:
            errs.append(f"no `impl {trait} for {typ}` ({site})")
            continue
        for it in items:
            if not re.search(r"\b" + re.escape(it) + r"\b", body):
                errs.append(f"`impl {trait} for {typ}` lacks `{it}`")
        ref = "/root/reference"
        if os.path.isdir(ref):  # the citation itself, when the reference is at hand (not on the GPU box)
            f, a, b = REFERENCE_LINES[trait]
            lines = open(os.path.join(ref, f)).read().split("\n")[a - 1:b]
            if not any(re.search(r"\b" + trait + r"\b", l) for l in lines):
                errs.append(f"{f}:{a}-{b} does not mention {trait}")
    return errs


C_KIND = [(r"\*|\[", "ptr"), (r"\buint64_t\b", "u64"), (r"\buint32_t\b", "u32"), (r"\buint8_t\b", "u8"), (r"\bsize_t\b", "usize"),
          (r"\bint\b", "i32"), (r"\bfloat\b", "f32"), (r"\bvoid\b", "void")]
R_KIND = [(r"^\*(const|mut)\b", "ptr"), (r"^u64$", "u64"), (r"^u32$", "u32"), (r"^u8$", "u8"), (r"^usize$", "usize"), (r"^(c_int|i32)$", "i32"),
          (r"^f32$", "f32")]


def c_kind(decl):
    for pat, k in C_KIND:
        if re.search(pat, decl):
            return k
    return "?" + decl


def r_kind(t):
    t = t.strip()
    for pat, k in R_KIND:
        if re.search(pat, t):
            return k
    return "?" + t


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        if ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def c_prototypes():
    h = open(os.path.join(ROOT, "include", "pcdhip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:const\s+)?[a-z_0-9]+\s*\**)\s*(pcdhip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", h, flags=re.M | re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        a = [] if args.strip() in ("", "void") else [c_kind(x) for x in split_args(args)]
        protos[name] = (c_kind(ret) if "*" in ret or not ret.strip().startswith("void") else "void", a)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(pcdhip_[a-z0-9_]+)\s*;", h, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            # `const uint64_t *a, *b;` declares several fields of one base type
            base = re.match(r"((?:const\s+)?[a-z_0-9]+)\s*(.*)", decl, flags=re.S)
            for name in split_args(base.group(2)):
                fields.append(c_kind(base.group(1) + " " + name))
        structs[m.group(2)] = fields
    return protos, structs


def rust_externs():
    fns, structs = {}, {}
    for f in sorted(os.listdir(os.path.join(RUST, "src"))):
        if not f.endswith(".rs"):
            continue
        src = re.sub(r"//[^\n]*", "", open(os.path.join(RUST, "src", f)).read())
        for blk in re.finditer(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S):
            for m in re.finditer(r"fn\s+(pcdhip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk.group(1), flags=re.S):
                args = [r_kind(a.split(":", 1)[1]) for a in split_args(m.group(2)) if ":" in a]
                fns[m.group(1)] = (r_kind(m.group(3)) if m.group(3) else "void", args)
        for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct\s+(pcdhip_[a-z0-9_]+)\s*\{(.*?)\}", src, flags=re.S):
            fields = [r_kind(a.split(":", 1)[1]) for a in split_args(m.group(2)) if ":" in a]
            structs[m.group(1)] = fields
    return fns, structs


def check_ffi():
    errs = []
    cp, cs = c_prototypes()
    rf, rs = rust_externs()
    for name, (rret, rargs) in sorted(rf.items()):
        if name not in cp:
            errs.append(f"{name}: declared in rust/src, not in include/pcdhip.h")
            continue
        cret, cargs = cp[name]
        if (cret, cargs) != (rret, rargs):
            errs.append(f"{name}: C ({cret}; {cargs}) != Rust ({rret}; {rargs})")
    for name, fields in sorted(rs.items()):
        if fields == ["?[u8; 0]"] or fields == ["ptr"] and name not in cs:  # opaque handles
            continue
        if name not in cs:
            if all(f.startswith("?[u8") for f in fields):
                continue
            errs.append(f"struct {name}: no C typedef")
        elif cs[name] != fields:
            errs.append(f"struct {name}: C {cs[name]} != Rust {fields}")
    return errs, len(rf), len([n for n in rs if n in cs])


def hook_kinds(text, name):
    m = re.search(r"pub type " + name + r"\s*=\s*unsafe fn\s*\((.*?)\)\s*->\s*bool", text, flags=re.S)
    return [a.split(":", 1)[1].strip() for a in split_args(m.group(1))] if m else None


def fn_kinds(text, name):
    m = re.search(r"unsafe fn " + name + r"\s*\((.*?)\)\s*->\s*bool", text, flags=re.S)
    return [a.split(":", 1)[1].strip() for a in split_args(m.group(1))] if m else None


def check_hooks():
    errs = []
    s2 = open(os.path.join(RUST, "src", "s2.rs")).read()
    ec = open(os.path.join(RUST, "s2_patch", "ark-ec", "src", "msm", "hook.rs")).read()
    po = open(os.path.join(RUST, "s2_patch", "ark-poly", "src", "domain", "hook.rs")).read()
    for typ, text, fn in (("MsmHook", ec, "msm_hook"), ("FftHook", po, "fft_hook")):
        a, b = hook_kinds(text, typ), fn_kinds(s2, fn)
        if a is None or b is None or a != b:
            errs.append(f"{typ} {a} != s2::{fn} {b}")
    for reg in ("ark_ec::msm::hook::set_msm_hook(msm_hook)", "ark_poly::domain::hook::set_fft_hook(fft_hook)"):
        if reg not in s2:
            errs.append(f"s2::install does not call {reg}")
    return errs


ROUND3_MISTAKE = {"ark-ec": "fn route() { let _ = core::any::TypeId::of::<ark_mnt4_298::G1Affine>(); }"}


def run(verbose=True):
    errs = []
    g, undeclared = dependency_graph()
    cyc = find_cycle(g)
    if cyc:
        errs.append("dependency cycle: " + " -> ".join(cyc))
    if undeclared:
        errs.append(f"rust/src names crates that rust/Cargo.toml does not declare: {sorted(undeclared)}")
    g3, _ = dependency_graph(ROUND3_MISTAKE)
    if not find_cycle(g3):
        errs.append("self-test: the round-3 layout (curve types named inside ark-ec) was NOT reported as a cycle")
    errs += check_bounds()
    e, nf, ns = check_ffi()
    errs += e
    errs += check_hooks()
    if verbose:
        print(f"crates: {len(g)}, acyclic: {not cyc}; extern \"C\" functions checked: {nf}; repr(C) structs checked: {ns}; trait impls checked: {len(BOUNDS)}")
        for x in errs:
            print("FAIL:", x)
    return errs


if __name__ == "__main__":
    sys.exit(1 if run() else 0)
