"""Audit of the four-wave GEMM kernels' code objects (CPU only; hipcc cross-compiles): the hand-scheduled loop names physical
registers -- v40..v255, all 256 accumulator registers -- inside one asm statement, and the C++ epilogue reads the accumulators
back with v_accvgpr_read statements.  That is only sound while the COMPILER never touches an accumulator register itself and
never spills (cdna_hip_programming.md section 5.7 item 4).  For every gemm_w4_kernel instantiation: no VGPR spill, no scratch,
no v_accvgpr_* / a[...] operand outside ;;#ASMSTART ... ;;#ASMEND.  Exit code 1 on a finding."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-gpt_amd", "csrc")


def audit(verbose=True):
    findings = []
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + CSRC,
               "-save-temps", "-c", os.path.join(CSRC, "gemm_bf16.hip"), "-o", os.path.join(d, "g.o")]
        subprocess.run(cmd, cwd=d, check=True, capture_output=True)
        s = open(os.path.join(d, "gemm_bf16-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S*gemm_w4_kernel\S*)\n(.*?)\.wavefront_size", s, flags=re.S):
        body = m.group(2)
        meta[m.group(1)] = {k: int(re.search(k + r":\s+(\d+)", body).group(1))
                            for k in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".vgpr_count")}
    names = sorted(meta)
    if len(names) != 12:   # PLAIN / GATED / ROPE x 256- and 192-wide tiles, PLAIN / ROPE x 288-wide, PLAIN with a transposed W, and with both operands transposed, x 256 / 192
        findings.append(f"expected 12 gemm_w4_kernel instantiations, found {len(names)}")
    for name in names:
        i = s.index("\n" + name + ":")
        j = s.index(".Lfunc_end", i)
        inside, n_mfma = False, 0
        for line in s[i:j].split("\n"):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                inside = True
            elif t.startswith(";;#ASMEND"):
                inside = False
            elif not inside and not t.startswith(";") and re.search(r"accvgpr|\ba\[\d|\ba\d+\b", t):
                findings.append(f"{name}: compiler instruction touches an accumulator register: {t}")
            elif inside and t.startswith("v_mfma"):
                n_mfma += 1
        mt = meta[name]
        if mt[".vgpr_spill_count"] or mt[".private_segment_fixed_size"]:
            findings.append(f"{name}: spills / scratch: {mt}")
        if verbose:
            print(f"{name}: {mt} mfma-in-asm {n_mfma}")
    return findings


def audit_attn_p2(verbose=True):
    """The hand-scheduled attention bodies (gen/attn_p2_gen.py) keep O, m, l, alpha and S in v[72:191] from one asm statement
    to the next with nothing but clobber lists telling hipcc about it: sound only while NO compiler-generated instruction
    between VGPT_P2_INIT and VGPT_P2_EXPORT writes one of those registers, and nothing spills."""
    findings = []
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + CSRC,
               "-save-temps", "-c", os.path.join(CSRC, "attn_fwd.hip"), "-o", os.path.join(d, "a.o")]
        subprocess.run(cmd, cwd=d, check=True, capture_output=True)
        s = open(os.path.join(d, "attn_fwd-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    names = [m.group(1) for m in re.finditer(r"\.name:\s+(\S*attn_fwd_kernelILi96ELb1ELi4ELb1E\S*)", s)]
    if len(names) != 1:
        return [f"expected one hand-scheduled attention kernel, found {names}"]
    name = names[0]
    body = re.search(r"\.name:\s+" + re.escape(name) + r"\n(.*?)\.wavefront_size", s, flags=re.S).group(1)
    mt = {k: int(re.search(re.escape(k) + r":\s+(\d+)", body).group(1)) for k in (".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".vgpr_count")}
    # hipcc keeps 64 registers around the bodies and parks a few values that are live from the prologue to the epilogue of a
    # work item (trace stamps, list-building pointers) in scratch: tolerated OUTSIDE the tile loop only -- checked below: no
    # scratch access between the barrier in front of the first steady body and the last steady body
    i = s.index("\n" + name + ":")
    lines = s[i:s.index(".Lfunc_end", i)].split("\n")
    marks, inside_, n_ = [], False, 0
    for ln, t in enumerate(l.strip() for l in lines):
        if t.startswith(";;#ASMSTART"):
            inside_, n_ = True, 0
        elif t.startswith(";;#ASMEND"):
            inside_ = False
            if n_ == 24:
                marks.append(ln)
        elif inside_ and t.startswith("v_mfma"):
            n_ += 1
    if len(marks) != 4:
        findings.append(f"expected 4 steady bodies (24 MFMAs each), found {len(marks)}")
    else:
        start = max(ln for ln, l in enumerate(lines[:marks[0]]) if l.strip() == "s_barrier")
        hot = [l.strip() for l in lines[start:marks[-1]] if "scratch_" in l]
        if hot:
            findings.append(f"{name}: scratch access inside the tile loop: {hot[:3]}")
    reg = re.compile(r"v(\d+)$|v\[(\d+):(\d+)\]$")
    inside, blocks, cur = False, [], []
    state, n_checked, n_mfma = "before", 0, 0
    for t in (l.strip() for l in lines):
        if t.startswith(";;#ASMSTART"):
            inside, cur = True, []
        elif t.startswith(";;#ASMEND"):
            inside = False
            text = "\n".join(cur)
            if "v_mov_b32 v72, 0xff800000" in text:
                if state != "before":
                    findings.append("a second VGPT_P2_INIT block")
                state = "live"
            elif re.search(r"v_mov_b32 v\d+, v80\b", text):
                state = "after"
            elif "v_mfma" in text:
                n_mfma += text.count("v_mfma")
                if state != "live":
                    findings.append(f"a tile body outside INIT .. EXPORT (state {state})")
        elif inside:
            cur.append(t)
        elif state == "live" and t and not t.startswith((";", ".")):
            parts = t.split(None, 1)
            if len(parts) == 2 and not parts[0].startswith(("s_", "ds_write", "global_store", "scratch_store", "buffer_store")):
                m = reg.match(parts[1].split(",")[0].strip())
                if m:
                    lo = int(m.group(1) if m.group(1) else m.group(2))
                    hi = lo if m.group(1) else int(m.group(3))
                    n_checked += 1
                    if hi >= 72 and lo <= 191:
                        findings.append(f"{name}: compiler instruction writes a state register between the bodies: {t}")
    if state != "after":
        findings.append(f"INIT / EXPORT blocks not found in order (state {state})")
    if verbose:
        print(f"{name}: {mt} mfma-in-asm {n_mfma}, {n_checked} compiler vector writes between INIT and EXPORT checked")
    return findings


if __name__ == "__main__":
    f = audit() + audit_attn_p2()
    for x in f:
        print("FINDING:", x)
    print("audit", "FAILED" if f else "ok")
    sys.exit(1 if f else 0)
