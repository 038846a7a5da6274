"""Developer tool: instruction mix of the device code of one group instantiation (no GPU needed).

    python tools/isa_mix.py 0          # MNT4-298 G1      python tools/isa_mix.py 4          # MNT4-753 G1
    python tools/isa_mix.py 0 blocks   # + the basic blocks of msm_accumulate_kernel<G, true> with >= 40 instructions

Compiles pcd_amd/csrc/inst_group.hip with -S for gfx950 and counts, per function / kernel, all instructions, the multiply-adds
(v_mad_u64_u32 + v_mad_i64_i32), scratch accesses and register moves; prints the register / scratch budget of the accumulate
kernel.  The fraction mads / instructions bounds `roofline_int.frac` from above: every VALU instruction of these kernels issues at
about the same rate as a v_mad_u64_u32 (profiles/r01_k0_int_rates.txt, r02_k1_mad_chain.txt).
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    idx = sys.argv[1] if len(sys.argv) > 1 else "0"
    blocks = len(sys.argv) > 2 and sys.argv[2] == "blocks"
    out = os.path.join(tempfile.gettempdir(), f"pcd_group{idx}.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-DPCD_GROUP_IDX={idx}", "--cuda-device-only", "-S",
                    os.path.join(ROOT, "pcd_amd/csrc/inst_group.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    body, meta, cur = {}, {}, None
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1); body[cur] = []; continue
        if line.startswith(".Lfunc_end"):
            cur = None; continue
        if cur is not None:
            body[cur].append(line)
        m = re.match(r"\s*\.amdhsa_kernel (\S+)", line)
        if m:
            kern = m.group(1); meta[kern] = {}
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|accum_offset) (\d+)", line)
        if m and meta:
            meta[kern][m.group(1)] = int(m.group(2))

    def demangle(s):
        return subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip().replace("pcd::", "")

    def mix(lines):
        c = collections.Counter()
        for l in lines:
            if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
                c[l.split()[0]] += 1
        return c

    def row(c):
        tot = sum(c.values()); mads = c["v_mad_u64_u32"] + c["v_mad_i64_i32"]
        scr = sum(v for k, v in c.items() if k.startswith("scratch_"))
        mov = sum(v for k, v in c.items() if k.startswith("v_mov") or k.startswith("v_accvgpr"))
        return tot, mads, scr, mov

    print(f"{'instr':>7s} {'mads':>6s} {'frac':>5s} {'scratch':>7s} {'moves':>6s}  function")
    for f, b in body.items():
        tot, mads, scr, mov = row(mix(b))
        if tot < 300:
            continue
        print(f"{tot:7d} {mads:6d} {mads / tot:5.2f} {scr:7d} {mov:6d}  {demangle(f)[:110]}")
    for k, v in meta.items():
        if "msm_accumulate_kernel" in k:
            print(f"{demangle(k)[:80]}: registers {v.get('next_free_vgpr')} (AGPRs from {v.get('accum_offset')}), "
                  f"scratch {v.get('private_segment_fixed_size')} B per lane")
    if blocks:
        for f, b in body.items():
            if "msm_accumulate_kernel" in f and "Lb1EEEv" in f:
                name, cur_lines = "entry", []
                print("basic blocks of", demangle(f)[:80])
                for l in b + [".LBBend_0:"]:
                    m = re.match(r"^(\.LBB\w+):", l)
                    if m:
                        tot, mads, scr, mov = row(mix(cur_lines))
                        if tot >= 40:
                            print(f"  {name:12s} instr={tot:5d} mads={mads:5d} scratch={scr:4d} moves={mov:4d}")
                        name, cur_lines = m.group(1), []
                    else:
                        cur_lines.append(l)


if __name__ == "__main__":
    main()
