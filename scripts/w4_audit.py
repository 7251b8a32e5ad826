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
    if len(names) != 10:   # PLAIN / GATED / ROPE x 256- and 192-wide tiles, PLAIN / ROPE x 288-wide, PLAIN with a transposed W x 256 / 192
        findings.append(f"expected 10 gemm_w4_kernel instantiations, found {len(names)}")
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


if __name__ == "__main__":
    f = audit()
    for x in f:
        print("FINDING:", x)
    print("audit", "FAILED" if f else "ok")
    sys.exit(1 if f else 0)
