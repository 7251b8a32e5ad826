#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the four-wave bf16 GEMM (csrc/gemm_w4_loop.inc, included by gemm_bf16.hip).

One workgroup = 4 waves (one per SIMD), tile 256 x (NI * 32) x 64, wave tile 128 x (NI * 16): MI = 8 A sub-tiles x NI W
sub-tiles of v_mfma_f32_16x16x32_bf16, all NI * 32 accumulator registers in AGPRs.  Operands go global -> VGPR
(buffer_load_dwordx4) -> LDS (ds_write_b128, the XOR-swizzled row images of gemm_bf16.hip) -> fragments (ds_read_b128),
with every wait counted by hand:

  iteration t = k-tile t, two k-steps of NI "steps" (step = one W sub-tile against the eight A sub-tiles = 8 MFMAs);
  k-step 0 multiplies fragment set 0 (A0, W0) and meanwhile reads set 1 (k-step 1 of the same tile, LDS buffer t & 1);
  ONE barrier; k-step 1 multiplies set 1 and reads set 0 of tile t + 1 (buffer (t + 1) & 1);
  staging: the interval between two barriers (k-step 1 of tile t - 1, k-step 0 of tile t) writes the pieces of tile t + 1
  into buffer (t + 1) & 1 -- free since the barrier of tile t - 1, first read behind the barrier of tile t -- one piece per
  step: s_waitcnt vmcnt(P - 1) (its fetch was issued a whole tile ago), ds_write_b128, and the fetch of the same piece of
  the tile after into the same registers.

The text is emitted as ONE asm statement (prologue, loop, drain); the C++ around it computes the tile's addresses and runs
the epilogue from the AGPRs.  `python gemm_w4_gen.py > ../gemm_w4_loop.inc`.
"""
import sys

# ---- physical registers (the asm statement's clobber list covers them; the compiler keeps v0..v39 / the low SGPRs) ----
V_R = 40          # staging registers: P pieces x 4
V_A = (104, 136)  # A fragments, set 0 / 1: 8 x 4 each
V_W = (168, 200)  # W fragments, set 0 / 1: up to 8 x 4 each
V_RDA = (232, 233)  # LDS read address, A, k-step 0 / 1
V_RDW = (234, 235)
V_WRA, V_WRW = 236, 237
V_OA, V_OW = 240, 248   # per-piece, per-lane source byte offsets (A: 8, W: up to 8): row clamped per lane, chunk swizzled
V_LAST = 255
S_KLOAD, S_KLAST, S_CNT = 56, 57, 58
S_FIRST, S_LAST = 36, 73
S_PIECE, S_TMP, S_DELTA = 36, 59, 53   # NI = 9: s36..s52 piece offsets, s53..s55 buffer deltas
S_KLOADW, S_KLASTW = 70, 71            # transposed W: its own k offset (64 rows per k-tile)
S_KLOADA, S_KLASTA = 72, 73            # transposed A (dW = dY^T X: both operands [K][..]): likewise
# diagnostics (python gemm_w4_gen.py --debug N; results are garbage unless N == 16): 1 = no global fetches in the loop,
# 2 = no LDS writes, 4 = no fragment reads, 8 = no barrier, 16 = s_memtime / s_memrealtime stamps around the loop and around
# every barrier (outputs %[cyc], %[rt], %[bar]: loop cycles, loop time in 10-ns ticks, cycles spent at the barriers)
DEBUG = 0
# schedule options under test (python gemm_w4_gen.py --split / --stagger N):
#   SPLIT: a piece goes to LDS as two ds_write_b64 in two MFMA gaps instead of one ds_write_b128 (13 issue cycles: more than
#          the 8 a gap leaves); STAGGER: waves 2, 3 idle N x 16 cycles behind every barrier, so that the two waves that share
#          a half of the LDS store path no longer issue their writes in the same cycle
SPLIT = False
STAGGER = 0
EARLY = True   # prologue: tile 1's first half requested beside tile 0 (--no-early: behind it, the first version)
TAIL = 1      # trailing steps of a k-step that carry MFMAs only (--tail N): the LDS queue drains under them, in front of the barrier
BUF_XOR = 0x8000


def vr(base, n=4):
    return f"v[{base}:{base + n - 1}]"


class Gen:
    def __init__(self, NI, wtr=False, atr=False):
        self.NI = NI
        self.wtr = wtr                    # W stored [K][N] (dX = dY W): fragments by ds_read_b64_tr_b16
        self.atr = atr                    # A stored [K][M] too (dW = dY^T X): its image and fragments as W's
        assert wtr or not atr
        self.PA, self.PW = 8, (8 if wtr else NI)   # pieces per wave and k-tile (a transposed W tile is a 256-column image)
        self.P = self.PA + self.PW
        if wtr:
            # transposed W: 64 reduction rows x 256 columns (512-byte rows, 32-byte units XOR-swizzled as in the eight-wave
            # kernel), a fragment = two ds_read_b64_tr_b16 at a per-lane address that is lane-dependent in the sub-tile index,
            # so one address VGPR per (fragment set, sub-tile) -- 2 NI of them -- and no room for register copies: every
            # address / offset is an asm operand used in place
            assert NI <= 8
            self.R = 64
            self.A = (128, 160)
            self.W = (192, 224)
            self.RDA, self.RDW = ("%[rdA0]", "%[rdA1]"), (None, None)
            self.WRA, self.WRW = "%[wrA]", "%[wrW]"
            self.sgpr_pieces = False
        elif NI <= 8:
            # physical registers; the operands are copied in by the prologue
            self.R, self.A, self.W = V_R, V_A, V_W
            self.RDA, self.RDW = tuple(f"v{x}" for x in V_RDA), tuple(f"v{x}" for x in V_RDW)
            self.WRA, self.WRW = f"v{V_WRA}", f"v{V_WRW}"
            self.sgpr_pieces = False
        else:
            # NI = 9 (256 x 288 tiles): 288 accumulator registers -- sub-tile column 8 lives in VGPRs -- leave no room for
            # copies: the LDS addresses are read-write operands used in place, the source offsets are one VGPR per operand
            # plus one SGPR per piece (no per-lane row clamp: whole tiles only), the LDS buffers are toggled by adding +-D
            self.R = 20
            self.A = (88, 120)
            self.W = (152, 188)
            self.ACC8 = 224
            self.RDA, self.RDW = ("%[rdA0]", "%[rdA1]"), ("%[rdW0]", "%[rdW1]")
            self.WRA, self.WRW = "%[wrA]", "%[wrW]"
            self.sgpr_pieces = True
        self.lines = []
        self.stash = {}                   # prologue only: piece -> registers that stand in for its staging registers
        self.lgkm = []                    # tags of issued LGKM operations, program order
        self.done = 0                     # lgkm[:done] are known complete
        self.waits = []                   # (iteration, position, text) of the counted waits, for the periodicity check
        # staging order: A and W pieces interleaved so that both operands' fetches are spread over the interval; the first
        # half is written in k-step 1 (of the tile before), the second half in k-step 0
        order = []
        for k in range(max(self.PA, self.PW)):
            if k < self.PA: order.append(k)
            if k < self.PW: order.append(self.PA + k)
        self.first, self.second = order[:self.P // 2], order[self.P // 2:]

    def emit(self, s):
        self.lines.append(s)

    # -- LGKM scoreboard --
    def issue(self, tag):
        self.lgkm.append(tag)

    def need(self, tags, it=None):
        idx = -1
        for t in tags:
            for k in range(len(self.lgkm) - 1, -1, -1):
                if self.lgkm[k] == t:
                    idx = max(idx, k)
                    break
            else:
                raise RuntimeError(f"fragment {t} was never requested")
        if idx < self.done:
            return
        n = len(self.lgkm) - 1 - idx
        n = min(n, 15)
        self.emit(f"s_waitcnt lgkmcnt({n})")
        self.waits.append((it, len(self.lines), n))
        self.done = len(self.lgkm) - n

    def drain(self):
        self.emit("s_waitcnt lgkmcnt(0)")
        self.done = len(self.lgkm)

    # -- pieces --
    def piece_regs(self, p):
        return self.stash.get(p, self.R + 4 * p)

    def load_piece(self, p, in_loop=True):
        if in_loop and (DEBUG & 1):
            return
        if self.sgpr_pieces:
            srd, vo = ("%[srdA]", "%[va]") if p < self.PA else ("%[srdW]", "%[vw]")
            self.emit(f"s_add_u32 s{S_TMP}, s{S_PIECE + p}, s{S_KLOAD}")
            self.emit(f"buffer_load_dwordx4 {vr(self.piece_regs(p))}, {vo}, {srd}, s{S_TMP} offen")
            return
        if self.wtr:
            if p < self.PA:
                self.emit(f"buffer_load_dwordx4 {vr(self.piece_regs(p))}, %[oa{p}], %[srdA], s{S_KLOADA if self.atr else S_KLOAD} offen")
            else:
                self.emit(f"buffer_load_dwordx4 {vr(self.piece_regs(p))}, %[ow{p - self.PA}], %[srdW], s{S_KLOADW} offen")
            return
        srd, vo = ("%[srdA]", V_OA + p) if p < self.PA else ("%[srdW]", V_OW + p - self.PA)
        self.emit(f"buffer_load_dwordx4 {vr(self.piece_regs(p))}, v{vo}, {srd}, s{S_KLOAD} offen")

    def toggle(self, regs, group):
        """Move LDS address registers to the other staging buffer."""
        if not self.sgpr_pieces:
            for r_ in regs:
                self.emit(f"v_xor_b32 {r_}, 0x{BUF_XOR:x}, {r_}")
            return
        sd = S_DELTA + group
        for r_ in regs:
            self.emit(f"v_add_u32 {r_}, s{sd}, {r_}")
        self.emit(f"s_sub_u32 s{sd}, 0, s{sd}")

    def write_piece(self, p, in_loop=True):
        if in_loop and (DEBUG & 2):
            return
        addr, q = (self.WRA, p) if p < self.PA else (self.WRW, p - self.PA)
        self.emit(f"ds_write_b128 {addr}, {vr(self.piece_regs(p))} offset:{q * 1024}")
        self.issue(("wr", p))

    def write_half(self, p, h, in_loop=True):
        if in_loop and (DEBUG & 2):
            return
        addr, q = (self.WRA, p) if p < self.PA else (self.WRW, p - self.PA)
        self.emit(f"ds_write_b64 {addr}, {vr(self.piece_regs(p) + 2 * h, 2)} offset:{q * 1024 + 8 * h}")
        self.issue(("wr", p, h))

    # -- fragments --
    def read_A(self, s, j, in_loop=True):
        if in_loop and (DEBUG & 4):
            self.issue(("A", s, j)); self.done = len(self.lgkm); return
        if self.atr:
            a = self.A[s] + 4 * j
            self.emit(f"ds_read_b64_tr_b16 {vr(a, 2)}, %[ra{s}_{j}] offset:{s * 32 * 512}")
            self.issue(("Alo", s, j))
            self.emit(f"ds_read_b64_tr_b16 {vr(a + 2, 2)}, %[ra{s}_{j}] offset:{s * 32 * 512 + 4 * 512}")
            self.issue(("A", s, j))
            return
        self.emit(f"ds_read_b128 {vr(self.A[s] + 4 * j)}, {self.RDA[s]} offset:{j * 2048}")
        self.issue(("A", s, j))

    def read_W(self, s, i, in_loop=True):
        if in_loop and (DEBUG & 4):
            self.issue(("W", s, i)); self.done = len(self.lgkm); return
        if self.wtr:
            # set s = k-step s of the tile: rows 32 s .. of the [64][256] image (512-byte rows); the upper half 4 rows on
            w = self.W[s] + 4 * i
            self.emit(f"ds_read_b64_tr_b16 {vr(w, 2)}, %[rw{s}_{i}] offset:{s * 32 * 512}")
            self.issue(("Wlo", s, i))
            self.emit(f"ds_read_b64_tr_b16 {vr(w + 2, 2)}, %[rw{s}_{i}] offset:{s * 32 * 512 + 4 * 512}")
            self.issue(("W", s, i))
            return
        self.emit(f"ds_read_b128 {vr(self.W[s] + 4 * i)}, {self.RDW[s]} offset:{i * 2048}")
        self.issue(("W", s, i))

    def mfma(self, s, i, j, it):
        self.need([("A", s, j), ("W", s, i)], it)
        if i < 8:
            a = (i * 8 + j) * 4
            acc = f"a[{a}:{a + 3}]"
        else:
            acc = vr(self.ACC8 + 4 * j)          # sub-tile column 8 of the 288-wide tile accumulates in VGPRs
        self.emit(f"v_mfma_f32_16x16x32_bf16 {acc}, {vr(self.W[s] + 4 * i)}, {vr(self.A[s] + 4 * j)}, {acc}")

    def kstep(self, s, pieces, it):
        """NI steps of 8 MFMAs on fragment set s; meanwhile: the reads of set s ^ 1, the given staging pieces, and the
        address toggles.  Nothing but MFMAs in the last step (the barrier / the loop edge follows)."""
        NI = self.NI
        o = s ^ 1
        reads = [("A", j) for j in range(8)] + [("W", i) for i in range(NI)]
        nsteps = NI - TAIL                                 # steps that carry side work
        def spread(n):                                     # n items over nsteps steps, front-loaded
            base, extra = divmod(n, nsteps)
            return [base + (1 if k < extra else 0) for k in range(nsteps)]
        rd_per, pc_per = spread(len(reads)), spread(len(pieces))
        ri = pi = 0
        for i in range(NI):
            side = []                                      # (slot after MFMA j, callable)
            if i < nsteps:
                slots_r = list(range(rd_per[i]))
                for k in range(rd_per[i]):
                    kind, x = reads[ri]; ri += 1
                    side.append((slots_r[k], (lambda kind=kind, x=x: self.read_A(o, x) if kind == "A" else self.read_W(o, x))))
                slot_w = [3, 5]
                slot_l = [4, 6]
                for k in range(pc_per[i]):
                    p = pieces[pi]; pi += 1
                    def wr(p=p):
                        if not (DEBUG & 1):
                            self.emit(f"s_waitcnt vmcnt({self.P - 1})")
                        if SPLIT:
                            self.write_half(p, 0)
                        else:
                            self.write_piece(p)
                    if rd_per[i] > 3 or pc_per[i] > 2:   # crowded step (--tail > 1): reads first, then pieces two gaps apart
                        s0 = min(rd_per[i] + 2 * k, 7)
                        side.append((s0, wr))
                        side.append((min(s0 + 1, 7), (lambda p=p: self.load_piece(p))))
                    elif SPLIT:   # reads in gaps 0..2; piece 0: halves in gaps 3, 4, fetch in 5; piece 1: 5 (after the fetch), 6, fetch in 7
                        s0 = (3, 5)[k] if k < 2 else 7
                        side.append((s0, wr))
                        side.append((min(s0 + 1, 7), (lambda p=p: self.write_half(p, 1))))
                        side.append((min(s0 + 2, 7), (lambda p=p: self.load_piece(p))))
                    else:
                        side.append((slot_w[k] if k < 2 else 7, wr))
                        side.append((slot_l[k] if k < 2 else 7, (lambda p=p: self.load_piece(p))))
            if i == NI - 1:
                # address toggles (VALU, no memory operation): the read addresses of the set just requested from
                def tog():
                    if self.atr:
                        self.toggle(tuple(f"%[ra{o}_{j_}]" for j_ in range(8)) + tuple(f"%[rw{o}_{i_}]" for i_ in range(NI)), o)
                    elif self.wtr:
                        self.toggle((self.RDA[o],) + tuple(f"%[rw{o}_{i_}]" for i_ in range(NI)), o)
                    else:
                        self.toggle((self.RDA[o], self.RDW[o]), o)
                side.append((1, tog))
            for j in range(8):
                self.mfma(s, i, j, it)
                for slot, fn in side:
                    if slot == j:
                        fn()
        assert ri == len(reads) and pi == len(pieces)

    def advance_k(self):
        """The fetches issued from here on belong to the next k-tile (clamped to the last one)."""
        self.emit(f"s_add_u32 s{S_KLOAD}, s{S_KLOAD}, 128")
        self.emit(f"s_min_u32 s{S_KLOAD}, s{S_KLOAD}, s{S_KLAST}")
        if self.wtr:      # a transposed W advances by 64 ROWS per k-tile
            self.emit(f"s_add_u32 s{S_KLOADW}, s{S_KLOADW}, %[wstep]")
            self.emit(f"s_min_u32 s{S_KLOADW}, s{S_KLOADW}, s{S_KLASTW}")
        if self.atr:
            self.emit(f"s_add_u32 s{S_KLOADA}, s{S_KLOADA}, %[astep]")
            self.emit(f"s_min_u32 s{S_KLOADA}, s{S_KLOADA}, s{S_KLASTA}")

    def body(self, it):
        P, PA, PW = self.P, self.PA, self.PW
        first, second = self.first, self.second
        # k-step 0: set 0; reads set 1 of this tile; writes the second half of tile t + 1, refetches it for tile t + 2
        self.kstep(0, second, it)
        self.drain()
        if DEBUG & 16:
            self.emit("s_memtime s[60:61]")
            self.emit("s_waitcnt lgkmcnt(0)")
        if not (DEBUG & 8):
            self.emit("s_barrier")
        if STAGGER:
            self.emit("s_bitcmp1_b32 %[wv], 1")
            self.emit("s_cbranch_scc0 2f")
            for _ in range(STAGGER):
                self.emit("s_nop 15")
            self.emit("2:")
        if DEBUG & 16:
            self.emit("s_memtime s[62:63]")
            self.emit("s_waitcnt lgkmcnt(0)")
            self.emit("s_sub_u32 s62, s62, s60")
            self.emit("s_add_u32 s64, s64, s62")
        self.toggle((self.WRA, self.WRW), 2)
        self.advance_k()
        self.emit(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
        # k-step 1: set 1; reads set 0 of tile t + 1; writes the first half of tile t + 2, refetches it for tile t + 3
        self.kstep(1, first, it)

    def generate(self):
        NI, P, PA, PW = self.NI, self.P, self.PA, self.PW
        e = self.emit
        lab = f"%="
        # ---- prologue ----
        if self.wtr:
            pass                                           # every address / offset is an operand used in place
        elif not self.sgpr_pieces:
            e(f"v_mov_b32 v{V_RDA[0]}, %[rdA]")
            e(f"v_xor_b32 v{V_RDA[1]}, 64, %[rdA]")
            e(f"v_mov_b32 v{V_RDW[0]}, %[rdW]")
            e(f"v_xor_b32 v{V_RDW[1]}, 64, %[rdW]")
            e(f"v_mov_b32 v{V_WRA}, %[wrA]")
            e(f"v_mov_b32 v{V_WRW}, %[wrW]")
            for p in range(PA):
                e(f"v_mov_b32 v{V_OA + p}, %[oa{p}]")
            for p in range(PW):
                e(f"v_mov_b32 v{V_OW + p}, %[ow{p}]")
        else:
            for p in range(P):                             # lane p of %[tab]: byte offset of piece p (A: 0..7, W: 8..16)
                e(f"v_readlane_b32 s{S_PIECE + p}, %[tab], {p}")
            for g_ in range(3):                            # +D: the step from staging buffer 0 to buffer 1
                e(f"s_mov_b32 s{S_DELTA + g_}, %[bufd]")
            e("s_nop 4")
        e(f"s_mov_b32 s{S_KLOAD}, 0")
        e(f"s_sub_u32 s{S_KLAST}, %[nk], 1")
        if self.wtr:
            e(f"s_mov_b32 s{S_KLOADW}, 0")
            e(f"s_mul_i32 s{S_KLASTW}, s{S_KLAST}, %[wstep]")
        if self.atr:
            e(f"s_mov_b32 s{S_KLOADA}, 0")
            e(f"s_mul_i32 s{S_KLASTA}, s{S_KLAST}, %[astep]")
        e(f"s_lshl_b32 s{S_KLAST}, s{S_KLAST}, 7")
        e(f"s_mov_b32 s{S_CNT}, %[nk]")
        first, second = self.first, self.second
        order = first + second
        for p in order:                                    # tile 0
            self.load_piece(p, False)
        if EARLY:
            # the first half of tile 1 is requested right behind tile 0, into the fragment registers of set 1 (free until the
            # fragment reads below): its latency runs beside tile 0's instead of behind it -- one global round trip less in
            # front of every tile's loop (one workgroup per CU: nothing else covers it)
            self.advance_k()
            slots = [self.A[1] + 4 * i for i in range(8)] + [self.W[1] + 4 * i for i in range(min(NI, 8))]
            for idx, p in enumerate(first):
                self.stash[p] = slots[idx]
                self.load_piece(p, False)
            self.stash = {}
        for a in range(0, min(NI, 8) * 32, 1):             # accumulators = 0 while the fetches fly
            e(f"v_accvgpr_write_b32 a{a}, 0")
        if NI > 8:
            for a in range(32):
                e(f"v_mov_b32 v{self.ACC8 + a}, 0")
        e(f"s_waitcnt vmcnt({len(first) if EARLY else 0})")
        for p in order:
            self.write_piece(p, False)
        if EARLY:
            for p in second:                               # second half of tile 1: in flight in its staging registers, as in the loop
                self.load_piece(p, False)
        else:
            self.advance_k()
            for p in order:                                # tile 1 (fetch order = the loop's consumption order)
                self.load_piece(p, False)
        self.toggle((self.WRA, self.WRW), 2)
        self.advance_k()
        for idx, p in enumerate(first):                    # "k-step 1 of tile -1": first half of tile 1 -> buffer 1, refetch for tile 2
            e(f"s_waitcnt vmcnt({P - 1})")
            if EARLY:
                self.stash[p] = slots[idx]
            self.write_piece(p, False)
            self.stash = {}
            self.load_piece(p, False)
        self.drain()
        e("s_barrier")
        for j in range(8):
            self.read_A(0, j, False)
            self.read_A(1, j, False)                       # (set 1 too: a diagnostics build that skips the reads still has data)
        for i in range(NI):
            self.read_W(0, i, False)
            self.read_W(1, i, False)
        # the read addresses of set 0 now point at the buffer of tile 1 (they are toggled at the end of every k-step 1 ... see kstep)
        if self.atr:
            self.toggle(tuple(f"%[ra0_{j_}]" for j_ in range(8)) + tuple(f"%[rw0_{i_}]" for i_ in range(NI)), 0)
        elif self.wtr:
            self.toggle((self.RDA[0],) + tuple(f"%[rw0_{i_}]" for i_ in range(NI)), 0)
        else:
            self.toggle((self.RDA[0], self.RDW[0]), 0)
        self.drain()
        if DEBUG & 16:
            e("s_mov_b32 s64, 0")
            e("s_memtime s[66:67]")
            e("s_memrealtime s[68:69]")
            e("s_waitcnt lgkmcnt(0)")
        # ---- loop: the body is generated three times on the scoreboard; the text of the last two must agree ----
        texts = []
        for it in range(3):
            n0 = len(self.lines)
            self.body(it)
            texts.append(self.lines[n0:])
            del self.lines[n0:]
        assert texts[1] == texts[2], "the counted waits are not periodic"
        e(f"1:")
        self.lines.extend(texts[2])
        e(f"s_cmp_lg_u32 s{S_CNT}, 0")
        e(f"s_cbranch_scc1 1b")
        # ---- drain ----
        e("s_waitcnt vmcnt(0) lgkmcnt(0)")
        if DEBUG & 16:
            e("s_memtime s[60:61]")
            e("s_memrealtime s[62:63]")
            e("s_waitcnt lgkmcnt(0)")
            e("s_sub_u32 %[cyc], s60, s66")
            e("s_sub_u32 %[rt], s62, s68")
            e("s_mov_b32 %[bar], s64")
        e("s_nop 15")
        e("s_nop 15")
        if NI > 8:
            # the VGPR accumulators (sub-tile column 8) leave through LDS: every wave's LDS traffic is over behind the barrier,
            # each lane writes and later reads back its own 8 x 16 bytes (%[dump] = its address)
            e("s_barrier")
            for j in range(8):
                e(f"ds_write_b128 %[dump], {vr(self.ACC8 + 4 * j)} offset:{j * 1024}")
            e("s_waitcnt lgkmcnt(0)")
        return self.lines


def c_string(lines):
    return "\n".join('    "' + l + '\\n\\t"' for l in lines)


def main():
    global DEBUG, SPLIT, STAGGER
    if "--debug" in sys.argv:
        DEBUG = int(sys.argv[sys.argv.index("--debug") + 1])
    SPLIT = "--split" in sys.argv
    global TAIL
    if "--tail" in sys.argv:
        TAIL = int(sys.argv[sys.argv.index("--tail") + 1])
    global EARLY
    if "--no-early" in sys.argv:
        EARLY = False
    if "--stagger" in sys.argv:
        STAGGER = int(sys.argv[sys.argv.index("--stagger") + 1])
    out = [f"// DIAGNOSTICS BUILD, debug = {DEBUG}" if DEBUG else "",
           "// GENERATED by gen/gemm_w4_gen.py -- do not edit; `make gemm_w4_loop.inc` regenerates it.",
           "// The hand-scheduled main loop of gemm_w4_kernel (gemm_bf16.hip): see the generator for the schedule.",
           ""]
    for NI, wtr, atr in ((8, False, False), (6, False, False), (9, False, False), (8, True, False), (6, True, False), (8, True, True),
                         (6, True, True)):
        g = Gen(NI, wtr, atr)
        lines = g.generate()
        n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
        out.append(f"// NI = {NI}{', A and W transposed' if atr else (', W transposed' if wtr else '')}: {len(lines)} lines, {n_mfma} MFMAs in the text (loop body {NI * 16})")
        out.append(f"#define VGPT_W4_ASM_NI{NI}{'_ATR' if atr else ('_WTR' if wtr else '')} \\")
        out.append(" \\\n".join('    "' + l + '\\n\\t"' for l in lines))
        out.append("")
    for name, v0 in (("VGPT_W4_CLOBBERS", V_R), ("VGPT_W4_CLOBBERS_NI9", 20), ("VGPT_W4_CLOBBERS_WTR", 64)):
        cl = [f'"v{i}"' for i in range(v0, V_LAST + 1)] + [f'"a{i}"' for i in range(256)] + \
             [f'"s{i}"' for i in range(S_FIRST, S_LAST + 1)] + ['"scc"', '"memory"']
        out.append(f"#define {name} \\")
        rows = [", ".join(cl[k:k + 16]) for k in range(0, len(cl), 16)]
        out.append(", \\\n".join("    " + r for r in rows))
        out.append("")
    sys.stdout.write("\n".join(out))


if __name__ == "__main__":
    main()
