#!/usr/bin/env python3
"""Generator of the hand-scheduled tile bodies of the attention forward (head dim 96, four waves, 32 query rows per wave).

Why: scripts/attn_issue_probe.py (profiles/r04_attn_issue_probe_v2_coissue_ladders.log) measured what a gfx950 SIMD issues
beside a v_mfma_f32_32x32x16_bf16 (32 cycles of the matrix pipe): up to six one-pass vector instructions (or three v_exp_f32)
for free, but a PACKED fp32 instruction (v_pk_fma / v_pk_mul / v_pk_add) waits for the MFMA in flight and blocks the next
one -- 53 cycles per group instead of 32.  The compiler-scheduled loop runs the two MFMA chains of a tile back to back and the
softmax (half of it packed) between them; two such waves on a SIMD overlap badly (1443 cycles per tile in the probe's program
order, ~1900 in the kernel, against 768 of MFMA + ~250 of packed work).  Here the tile is software-pipelined by hand:

  phase 1   S(t+1)^T = K(t+1) Q^T  (12 MFMAs)      beside  P(t) = exp2(x(t)) and its rounding to bf16   (v_exp / v_cvt_pk only)
  block     row sums (v_pk_add), l update, O *= alpha(t) when a row maximum moved (v_pk_mul)            (no MFMA in flight)
  phase 2   O^T += V(t)^T P(t)^T   (12 MFMAs)      beside  row maxima of S(t+1), m / alpha update, x(t+1) = S scale - m
                                                           (v_max3 / v_fma, one-pass only)

K fragments are read at the top of phase 1 into a 48-register pool; as the k-steps of QK^T retire their fragments the V
fragments of the P.V product take the registers over.  Arithmetic, operand layouts and summation order are those of the C++
tile body in attn_fwd.hip (same instructions per element: v_fma, v_exp, v_cvt_pk_bf16_f32, the row-sum association), so the
kernels agree bit for bit (tests/test_ops_gpu.py).  Control (tile list, barriers, LDS-DMA) stays in C++ around the bodies.

Emits attn_p2_loop.inc: VGPT_P2_PRO_<p>, VGPT_P2_STEADY_<p>, VGPT_P2_DRAIN_<p> for p = parity of the staging buffer that
holds the CURRENT tile (the S registers alternate with it), and the operand / clobber lists.
"""
import os
import sys

# ---- physical registers ------------------------------------------------------------------------------------------------
V_T = 64          # v[64:71] temporaries
V_M, V_L, V_ALPHA = 72, 73, 74   # running maximum, running sum, alpha of the current tile (v75: pad of the alpha pair)
V_MUSE = 76       # m_use of the tile whose x is being formed
V_MOVED = 77      # 1 in the lanes whose running maximum moved with the current tile (decides the rescale one body later; a
                  # scalar flag cannot cross the C++ between two bodies: hipcc treats every asm result as divergent)
V_O = 80          # O^T accumulators: 3 x 16
V_S = (128, 160)  # S^T / x / P~ of the tile in staging buffer 0 / 1: 2 x 16 each
V_P = 192         # P fragments (bf16 pairs): 4 x 4
V_F = 208         # fragment pool: 12 K fragments (4 registers) or 12 V fragments (2 + 2)
KROW, STAGE, KBYTES = 192, 24576, 12288


def S(p, kb, i):
    return V_S[p] + 16 * kb + i


class Sched:
    """instruction list with an LDS scoreboard (lgkmcnt is in order for LDS)"""
    def __init__(self):
        self.out, self.lds = [], []      # lds: tags of outstanding reads in issue order

    def emit(self, s):
        if (DIAG & 8) and s.startswith("v_exp_f32"):
            s = s.replace("v_exp_f32", "v_mov_b32")
        if (DIAG & 16) and s.startswith("v_mfma"):
            return
        if (DIAG & 4) and s.startswith("global_load_lds"):
            return
        self.out.append(s)

    def read(self, tag, s):
        if (DIAG & 1) and tag[0] == "v":
            return
        if (DIAG & 2) and tag[0] == "k":
            return
        self.out.append(s)
        self.lds.append(tag)

    def need(self, tags):
        """wait until every read in `tags` has landed"""
        last = max((i for i, t in enumerate(self.lds) if t in tags), default=-1)
        if last < 0:
            return
        allowed = len(self.lds) - 1 - last
        self.out.append(f"s_waitcnt lgkmcnt({min(allowed, 15)})")
        self.lds = self.lds[last + 1:] if allowed <= 15 else self.lds[len(self.lds) - 15:]


def k_read(sc, buf, j):
    """K fragment j = (k-step j >> 1, key half j & 1) of the tile in staging buffer `buf` -> pool[4 j ..)"""
    s, kb = j >> 1, j & 1
    off = buf * STAGE + kb * 32 * KROW + 64 * (s >> 1)
    sc.read(("k", j), f"ds_read_b128 v[{V_F + 4 * j}:{V_F + 4 * j + 3}], %[ka{s & 1}] offset:{off}")


def v_reads(sc, buf, j):
    """V fragment j = (key step t = j // 3, d block dt = j % 3): two transposed 8-byte reads -> pool[4 j ..)"""
    t, dt = j // 3, j % 3
    off = buf * STAGE + KBYTES + ((t >> 1) * 32 + (t & 1) * 16) * KROW + dt * 64
    sc.read(("v", j), f"ds_read_b64_tr_b16 v[{V_F + 4 * j}:{V_F + 4 * j + 1}], %[va] offset:{off}")
    sc.read(("v", j), f"ds_read_b64_tr_b16 v[{V_F + 4 * j + 2}:{V_F + 4 * j + 3}], %[va] offset:{off + 8 * KROW}")


def qk_mfma(p, j):
    s, kb = j >> 1, j & 1
    acc = f"v[{S(p, kb, 0)}:{S(p, kb, 15)}]"
    c = "0" if s == 0 else acc
    return f"v_mfma_f32_32x32x16_bf16 {acc}, v[{V_F + 4 * j}:{V_F + 4 * j + 3}], %[q{s}], {c}"


def pv_mfma(j):
    t, dt = j // 3, j % 3
    acc = f"v[{V_O + 16 * dt}:{V_O + 16 * dt + 15}]"
    return f"v_mfma_f32_32x32x16_bf16 {acc}, v[{V_F + 4 * j}:{V_F + 4 * j + 3}], v[{V_P + 4 * t}:{V_P + 4 * t + 3}], {acc}"


def exp_cvt_ops(p):
    """(cycles, text): P~ = exp2(x) in place, then the bf16 pairs; a v_cvt_pk sits at least two instructions behind the
    v_exp_f32 that feed it (trans-use hazard)"""
    regs = [S(p, kb, i) for kb in range(2) for i in range(16)]
    ex = [(8, f"v_exp_f32 v{r}, v{r}") for r in regs]
    cv = [(4, f"v_cvt_pk_bf16_f32 v{V_P + q}, v{regs[2 * q]}, v{regs[2 * q + 1]}") for q in range(16)]
    ops = ex[:4]
    for q in range(14):
        ops += [cv[q], ex[4 + 2 * q], ex[5 + 2 * q]]
    ops += [cv[14], (4, "s_nop 0"), cv[15]]
    return ops


def rowsum_block(p):
    """row sums in the C++ body's association: two packed chains over the register pairs (0,1),(4,5).. / (2,3),(6,7).. of
    both key halves, then (x + y) + (x + y), the partner half-wave's sum, l = l alpha + rs as one fma; O *= alpha behind a
    wave-uniform branch"""
    t0, t1 = V_T, V_T + 2          # chain registers (pairs)
    o = []
    first = [True, True]
    for kb in range(2):
        for i in range(0, 16, 2):
            c = (i >> 1) & 1
            dst = t0 if c == 0 else t1
            src = f"v[{S(p, kb, i)}:{S(p, kb, i + 1)}]"
            if first[c]:
                o.append(f"v_pk_mov_b32 v[{dst}:{dst + 1}], {src}, {src} op_sel:[0,1]")
                first[c] = False
            else:
                o.append(f"v_pk_add_f32 v[{dst}:{dst + 1}], {src}, v[{dst}:{dst + 1}]")
    o += [f"v_cmp_ne_u32 vcc, 0, v{V_MOVED}",
          f"v_add_f32 v{t0}, v{t0}, v{t0 + 1}", f"v_add_f32 v{t1}, v{t1}, v{t1 + 1}", f"v_add_f32 v{t0}, v{t0}, v{t1}",
          f"v_mov_b32 v{t1}, v{t0}", "s_nop 1", f"v_permlane32_swap_b32 v{t0}, v{t1}",
          f"v_add_f32 v{t0}, v{t0}, v{t1}", f"v_fmac_f32 v{t0}, v{V_L}, v{V_ALPHA}", f"v_mov_b32 v{V_L}, v{t0}",
          "s_cbranch_vccz 7f"]
    for i in range(24):
        o.append(f"v_pk_mul_f32 v[{V_O + 2 * i}:{V_O + 2 * i + 1}], v[{V_ALPHA}:{V_ALPHA + 1}], v[{V_O + 2 * i}:{V_O + 2 * i + 1}] op_sel_hi:[0,1]")
    o.append("7:")
    return o


def part1_ops(p, masked):
    """[mask of a tile the wave does not see in full,] row maxima of the raw scores of the tile in S[p], m / alpha / flag
    update, x = S scale - m_use (one-pass instructions).  Mask: score register i of key half kb is key 32 kb + (i & 3) +
    8 (i >> 2) + 4 h; %[mw<kb>] holds that half's 32 key bits of the lane's query row, shifted right by 4 h: the bit spread to
    0 / ~0 (v_bfe_i32), then (s & m) | (-inf & ~m) (v_bitop3_b32) -- the C++ body's two instructions"""
    a, b, t = V_T + 4, V_T + 5, V_T + 6
    o = []
    if masked:
        for kb in range(2):
            for i in range(16):
                tmp = V_T + (i & 3)
                o.append(f"v_bfe_i32 v{tmp}, %[mw{kb}], {(i & 3) + 8 * (i >> 2)}, 1")
                if i & 1:
                    for k in (i - 1, i):
                        o.append(f"v_bitop3_b32 v{S(p, kb, k)}, v{S(p, kb, k)}, %[ninf], v{V_T + (k & 3)} bitop3:0xe4")
    pre = o
    o = []
    for kb, acc in ((0, a), (1, b)):
        r = [S(p, kb, i) for i in range(16)]
        o.append(f"v_max3_f32 v{acc}, v{r[0]}, v{r[1]}, v{r[2]}")
        for i in range(3, 15, 2):
            o.append(f"v_max3_f32 v{acc}, v{acc}, v{r[i]}, v{r[i + 1]}")
        o.append(f"v_max_f32 v{acc}, v{acc}, v{r[15]}")
    # interleave the two chains (dependent v_max3 back to back would expose their latency)
    n = len(o) // 2
    o = [x for pair in zip(o[:n], o[n:]) for x in pair]
    o += [f"v_max_f32 v{a}, v{a}, v{b}", f"v_mov_b32 v{b}, v{a}", "s_nop 1", f"v_permlane32_swap_b32 v{a}, v{b}",
          f"v_max_f32 v{a}, v{a}, v{b}", f"v_mul_f32 v{a}, %[scale], v{a}", f"v_max_f32 v{a}, v{V_M}, v{a}",      # a = m_new
          f"v_cmp_neq_f32 vcc, v{a}, v{V_M}", "s_nop 1", f"v_cndmask_b32 v{V_MOVED}, 0, 1, vcc",
          f"v_cmp_neq_f32 vcc, 0xff800000, v{a}", "s_nop 1", f"v_cndmask_b32 v{V_MUSE}, 0, v{a}, vcc",
          f"v_sub_f32 v{t}, v{V_M}, v{V_MUSE}", f"v_exp_f32 v{V_ALPHA}, v{t}", f"v_mov_b32 v{V_M}, v{a}"]
    if PKFMA:
        # x = S scale - m_use two at a time: the diagnostics (P2_DIAG) show the kernel bound by the vector issue port the two
        # waves of a SIMD share, not by the matrix pipe -- a packed instruction stalls the MFMA in flight but halves the slots
        for kb in range(2):
            for i in range(0, 16, 2):
                r = S(p, kb, i)
                o.append(f"v_pk_fma_f32 v[{r}:{r + 1}], v[{r}:{r + 1}], %[scale2], v[{V_MUSE}:{V_MUSE + 1}] op_sel_hi:[1,1,0] neg_lo:[0,0,1] neg_hi:[0,0,1]")
    else:
        for kb in range(2):
            for i in range(16):
                o.append(f"v_fma_f32 v{S(p, kb, i)}, v{S(p, kb, i)}, %[scale], -v{V_MUSE}")
    return [(4, x) for x in pre + o]


DMA_V_EARLY = int(os.environ.get("P2_DMA_V_EARLY", 1))


def dma(op, j):
    """LDS-DMA piece j (of 3) of the K image of tile t+2 / the V image of tile t+1: 1 KiB, lane i's 16 bytes at base + offset
    land at M0 + 16 i.  Spread through the body: issued in one burst behind the barrier the four waves of a workgroup (and
    the partner workgroup's) queue at the texture unit -- ~900 cycles of a 2.8 k-cycle tile in the stamps of the first build."""
    o = []
    if j == 0:
        o.append(f"s_mov_b32 m0, %[{op}dst]")
    else:       # (K and V pieces interleave: M0 is formed from the operand's base every time)
        o.append(f"s_add_u32 m0, %[{op}dst], {j * 1024}")
    o += ["s_nop 0", f"global_load_lds_dwordx4 %[{op}o{j}], %[{op}base]"]
    return o


def fill(sc, queue, budget):
    while queue and budget >= queue[0][0]:
        c, t = queue.pop(0)
        sc.emit(t)
        budget -= c


SLOT = 24      # cycles of one-pass vector issue that fit beside one MFMA (32 - 8 of MFMA issue)


K_AHEAD = int(os.environ.get("P2_KAHEAD", 6))     # K fragments requested before the first MFMA; one more behind every MFMA
DIAG = int(os.environ.get("P2_DIAG", 0))   # diagnostics (results are WRONG): 1 no V fragment reads, 2 no K fragment reads, 4 no LDS-DMA, 8 v_exp_f32
                                           # replaced by v_mov_b32, 16 no MFMAs
PKFMA = int(os.environ.get("P2_PKFMA", 0))     # x = S scale - m as 16 v_pk_fma_f32 instead of 32 v_fma_f32
PREFILL = int(os.environ.get("P2_PREFILL", 48))   # cycles of exponentials issued while the first K fragments travel


def phase_qk(sc, p_next, buf_next, valu, with_dma=False):
    """QK^T of the tile in buffer buf_next into S[p_next]; `valu` (exp / cvt of the current tile) fills the slots.  The LDS
    traffic of a tile is split between the phases (K here, V beside P.V): with all 36 reads in this phase the four waves of a
    workgroup, in lockstep behind the barrier, asked the LDS for ~190 B/clk."""
    for j in range(K_AHEAD):
        k_read(sc, buf_next, j)
    fill(sc, valu, PREFILL)
    for j in range(12):
        sc.need({("k", j)})
        sc.emit(qk_mfma(p_next, j))
        budget = SLOT
        if j + K_AHEAD < 12:
            k_read(sc, buf_next, j + K_AHEAD)
            budget -= 4
        if with_dma and j in (1, 5, 9):
            for t in dma("k", (j - 1) // 4):
                sc.emit(t)
            budget -= 8
        if with_dma and DMA_V_EARLY and j in (3, 7, 11):
            # V(t+1) goes out in THIS phase too: issued beside P.V its pieces had ~400 cycles before the next body's
            # s_waitcnt vmcnt(0) -- a tile's worth of L2 latency was exposed at the top of every body
            for t in dma("v", (j - 3) // 4):
                sc.emit(t)
            budget -= 8
        fill(sc, valu, budget)
    while valu:
        sc.emit(valu.pop(0)[1])


def phase_pv(sc, buf_cur, valu, with_dma=False, first_requested=True):
    """P.V of the current tile; the V fragments of key step t + 1 are requested before the MFMAs of key step t (those of key
    step 0 by the caller, ahead of the row-sum block)"""
    if not first_requested:
        for j in range(3):
            v_reads(sc, buf_cur, j)
    for t in range(4):
        if t < 3:
            for j in range(3 * (t + 1), 3 * (t + 2)):
                v_reads(sc, buf_cur, j)
        for j in range(3 * t, 3 * t + 3):
            sc.need({("v", j)})
            sc.emit(pv_mfma(j))
            budget = SLOT - (8 if (t < 3 and j == 3 * t) else 0)
            if with_dma and not DMA_V_EARLY and j in (1, 5, 9):
                for x in dma("v", (j - 1) // 4):
                    sc.emit(x)
                budget -= 8
            fill(sc, valu, budget)
    while valu:
        sc.emit(valu.pop(0)[1])


def body(kind, p, masked=False):
    """p = staging buffer (and S register set) of the CURRENT tile; masked: the tile whose maxima this body forms (the first
    tile in `pro`, the NEXT one in `steady`) is not visible in full to this wave"""
    sc = Sched()
    if not int(os.environ.get("P2_NONOP", 0)):
        sc.emit("s_nop 3")
    if kind == "pro":          # QK^T of the first tile of a run, then its maxima: no tile before it
        phase_qk(sc, p, p, [])
        sc.emit("s_nop 7"); sc.emit("s_nop 3")          # MFMA result -> vector reader
        for c, t in part1_ops(p, masked):
            sc.emit(t)
    elif kind == "steady":
        sc.emit("s_mov_b32 %[m0keep], m0")
        phase_qk(sc, p ^ 1, p ^ 1, exp_cvt_ops(p), True)
        for j in range(3):
            v_reads(sc, p, j)
        for t in rowsum_block(p):
            sc.emit(t)
        phase_pv(sc, p, part1_ops(p ^ 1, masked), True)
        sc.emit("s_mov_b32 m0, %[m0keep]")
    else:                      # drain: the last tile of a run
        for j in range(3):
            v_reads(sc, p, j)
        for c, t in exp_cvt_ops(p):
            sc.emit(t)
        for t in rowsum_block(p):
            sc.emit(t)
        phase_pv(sc, p, [])
        sc.emit("s_nop 7"); sc.emit("s_nop 3")          # the C++ epilogue reads O next
    sc.emit("s_nop 1")
    return sc.out


def main():
    out = ["// GENERATED by gen/attn_p2_gen.py -- do not edit (tests/test_w4_audit.py regenerates and compares).",
           "// Hand-scheduled tile bodies of attn_fwd_kernel<96, true, 4, P2>; register map and schedule: see the generator."]
    for kind in ("pro", "steady", "drain"):
        for p in (0, 1):
            for masked in ((False,) if kind == "drain" else (False, True)):
                lines = body(kind, p, masked)
                out.append(f"#define VGPT_P2_{kind.upper()}_{p}{'_M' if masked else ''} \\")
                out += [f'    "{l}\\n" \\' for l in lines]
                out.append("")
    # The state (O, m, l, alpha, S) lives in the physical registers BETWEEN the bodies: every body clobbers v64..v255, no C++
    # variable is bound to them (hipcc spilled hundreds of registers around 7 pinned 16-register operands x 10 statements), and
    # scripts/w4_audit.py checks in the code object that no compiler instruction between VGPT_P2_INIT and VGPT_P2_EXPORT
    # writes one of them.
    init = [f"v_mov_b32 v{V_M}, 0xff800000", f"v_mov_b32 v{V_L}, 0", f"v_mov_b32 v{V_ALPHA}, 1.0", f"v_mov_b32 v{V_MOVED}, 0"]
    init += [f"v_mov_b32 v{V_O + i}, 0" for i in range(48)]
    out.append("#define VGPT_P2_INIT \\")
    out += [f'    "{l}\\n" \\' for l in init]
    out.append("")
    exp = [f"v_mov_b32 %{i}, v{V_O + i}" for i in range(48)] + [f"v_mov_b32 %48, v{V_M}", f"v_mov_b32 %49, v{V_L}"]
    out.append("#define VGPT_P2_EXPORT \\")
    out += [f'    "{l}\\n" \\' for l in exp]
    out.append("")
    clob = [f"v{i}" for i in range(V_T, 256)]
    out.append("#define VGPT_P2_CLOBBERS " + ", ".join(f'"{c}"' for c in clob) + ', "vcc", "scc", "memory"')
    print("\n".join(out))


if __name__ == "__main__":
    main()
