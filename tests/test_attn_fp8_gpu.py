"""MX-fp8 attention (video-gpt_amd/csrc/attn_fp8.hip; the "fp8 MFMA attention" option of cfg-5) through
the C ABI: vgpt_attn_fp8_quantize + vgpt_attn_fwd_plan_fp8.

Tolerance, stated separately from the bf16 path (whose bar is rel-L2 <= 1e-2 against the fp64 reference):
  * operands are OCP e4m3 (3 mantissa bits: relative rounding error <= 2^-4 per element, RMS ~ 2^-4 / sqrt(3) = 3.6e-2)
    in blocks of 32 sharing a power-of-two scale; products accumulate in fp32;
  * against the SAME attention evaluated in fp64 on the dequantised Q / K / V (what is left is the rounding of the
    probabilities, averaged over the visible keys): rel-L2 <= 3e-2;
  * against the fp64 reference on the unquantised bf16 inputs (the oracle's arithmetic): rel-L2 <= 8e-2 on unit-normal
    q / k / v -- the score error of fp8 Q.K^T (RMS ~ 3.6e-2 * sqrt(2) * |q.k| / sqrt(d) per score) moves the softmax weights.
The quantiser itself is checked exactly: every stored byte equals the e4m3 rounding of x / 2^e with the block's
E8M0 exponent e, in the record layout include/vgpt.h documents.
"""
import importlib
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BF = torch.bfloat16
D = 96
REC = 13 * 1024
KS_OFF, V8_OFF, VS_OFF = 6144, 6400, 12544


def g(seed):
    return torch.Generator("cpu").manual_seed(seed)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def e4m3_table():
    t = np.zeros(256, dtype=np.float64)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        if e == 15 and m == 7:
            v = np.nan
        t[b] = -v if s else v
    return t


def quantise_blocks(x):
    """numpy model of the kernel's block quantiser: x (..., 32) -> (dequantised values, E8M0 bytes).  Rounding to e4m3 is
    round-to-nearest-even on the 3-bit mantissa (values are <= 448 after scaling, so nothing saturates)."""
    amax = np.abs(x).max(axis=-1, keepdims=True)
    _, e = np.frexp(amax.astype(np.float32) * np.float32(1.0 / 448.0))   # the kernel's fp32 arithmetic
    e = np.where(amax > 0, e, 0)
    y = x * 2.0 ** (-e.astype(np.float64))
    tab = e4m3_table()
    pos = np.sort(tab[:127])                      # non-negative finite values, ascending
    a = np.abs(y)
    idx = np.clip(np.searchsorted(pos, a), 1, len(pos) - 1)
    lo, hi = pos[idx - 1], pos[idx]
    pick_hi = (a - lo > hi - a) | ((a - lo == hi - a) & ((idx % 2) == 0))   # ties to the even mantissa (index parity)
    q = np.where(pick_hi, hi, lo) * np.sign(y)
    return q * 2.0 ** e.astype(np.float64), (e + 127).astype(np.uint8)[..., 0]


def split_qkv(qkv, nh, nkv):
    B, L, _ = qkv.shape
    q = qkv[..., : nh * D].view(B, L, nh, D).transpose(1, 2)
    k = qkv[..., nh * D:(nh + nkv) * D].view(B, L, nkv, D).transpose(1, 2)
    v = qkv[..., (nh + nkv) * D:].view(B, L, nkv, D).transpose(1, 2)
    return q, k, v


def ref_attention(q, k, v, mask, scale):
    s = torch.matmul(q.double(), k.double().transpose(2, 3)) * scale
    s = s.masked_fill(~mask[:, None].bool(), float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v.double())


def random_block_mask(B, L, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros((B, L, L), dtype=np.uint8)
    for b in range(B):
        cuts = np.sort(rng.choice(np.arange(1, L), size=min(5, L - 1), replace=False))
        bounds = [0, *cuts.tolist(), L]
        for i in range(len(bounds) - 1):
            for j in range(i + 1):
                if rng.random() < 0.7 or i == j:
                    m[b, bounds[i]:bounds[i + 1], bounds[j]:bounds[j + 1]] = 1
        m[b] |= np.eye(L, dtype=np.uint8)
    return m


def test_quantiser_layout_and_rounding(ops):
    """Workspace bytes against the numpy model: Q8 / QS, and per 64-key tile K8 | KS | V8 (P^T operand order) | VS."""
    L_ = importlib.import_module("video-gpt_amd._lib")
    B, L, nh, nkv = 1, 150, 2, 1
    scale = 1 / math.sqrt(D)
    qkv = (torch.randn(B, L, (nh + 2 * nkv) * D, generator=g(5)) * torch.logspace(-2, 1.5, (nh + 2 * nkv) * D)).to(BF)
    dq = qkv.to(DEV)
    nbytes = int(L_.load().vgpt_attn_fp8_workspace_bytes(B, L, nh, nkv, D))
    nkt = (L + 63) // 64
    a256 = lambda n: (n + 255) // 256 * 256
    assert nbytes == a256(B * nh * L * D) + a256(B * nh * L * 4) + B * nkv * nkt * REC
    assert int(L_.load().vgpt_attn_fp8_workspace_bytes(B, L, nh, nkv, 128)) == -1
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    w = (nh + 2 * nkv) * D
    kq = dq.data_ptr() + nh * D * 2
    vq = kq + nkv * D * 2
    L_.call("vgpt_attn_fp8_quantize", dq.data_ptr(), kq, vq, ws.data_ptr(), B, L, 0, nh, nkv, D, L * w, D, w, L * w, D, w, L * w, D, w,
            float(scale), ops._stream())
    raw = ws.cpu().numpy()
    tab = e4m3_table()
    q, k, v = [t.float().numpy().astype(np.float64) for t in split_qkv(qkv, nh, nkv)]
    # Q (pre-multiplied by scale * log2 e in fp32, as the kernel does)
    qmul = np.float32(np.float32(scale) * np.float32(1.4426950408889634))
    qx = (q.astype(np.float32) * qmul).astype(np.float64).reshape(B, nh, L, 3, 32)
    want, sc = quantise_blocks(qx)
    q8 = raw[: B * nh * L * D].reshape(B, nh, L, 3, 32)
    qs = raw[a256(B * nh * L * D): a256(B * nh * L * D) + B * nh * L * 4].reshape(B, nh, L, 4)
    assert np.array_equal(qs[..., :3], sc)
    assert np.array_equal(tab[q8] * 2.0 ** (qs[..., :3, None].astype(np.float64) - 127), want)
    # K / V records
    recs = raw[a256(B * nh * L * D) + a256(B * nh * L * 4):].reshape(B, nkv, nkt, REC)
    kp = np.zeros((B, nkv, nkt * 64, D)); kp[:, :, :L] = k
    vp = np.zeros((B, nkv, nkt * 64, D)); vp[:, :, :L] = v
    for kt in range(nkt):
        rec = recs[0, 0, kt]
        want, sc = quantise_blocks(kp[0, 0, kt * 64:(kt + 1) * 64].reshape(64, 3, 32))
        ks = rec[KS_OFF: KS_OFF + 256].reshape(64, 4)
        assert np.array_equal(ks[:, :3], sc)
        k8 = rec[: 64 * D].reshape(64, 3, 32)
        assert np.array_equal(tab[k8] * 2.0 ** (ks[:, :3, None].astype(np.float64) - 127), want)
        vt = vp[0, 0, kt * 64:(kt + 1) * 64]                       # (64 keys, 96)
        blocks = vt.reshape(2, 32, 3, 32).transpose(2, 0, 3, 1)     # (dt, key half, d, 32 keys)
        want, sc = quantise_blocks(blocks)
        vs = rec[VS_OFF: VS_OFF + 192].reshape(3, 2, 32)
        assert np.array_equal(vs, sc)
        v8 = rec[V8_OFF: V8_OFF + 6144].reshape(3, 2, 32, 32)       # (dt, lane half h, d, byte)
        got = np.zeros((3, 2, 32, 32))                              # back to (dt, key half, d, key in half)
        for kb in range(2):
            for kk in range(32):
                h, j = (kk >> 2) & 1, kb * 16 + (kk & 3) + 4 * (kk >> 3)
                got[:, kb, :, kk] = tab[v8[:, h, :, j]] * 2.0 ** (vs[:, kb, :].astype(np.float64) - 127)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("B,L,nh,nkv,segs", [(1, 700, 4, 4, None), (2, 330, 4, 2, None), (1, 64, 2, 2, None), (1, 1, 1, 1, None),
                                             (1, 900, 3, 3, ((0, 300, 316), (0, 316, 900)))])
def test_attention_fp8_against_references(ops, B, L, nh, nkv, segs):
    m = random_block_mask(B, L, 3)
    m[:, : L // 3, L // 2:] = 0
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    qkv = torch.randn(B, L, (nh + 2 * nkv) * D, generator=g(13)).to(BF)
    scale = 1 / math.sqrt(D)
    q_start = segs[0][1] if segs else 0
    out = torch.full((B, L - q_start, nh * D), 7.0, dtype=BF, device=DEV)
    ops.attention_qkv_fp8(qkv.to(DEV), pm, nh, nkv, D, out=out, q_start=q_start, segments=segs)
    assert torch.isfinite(out).all()
    q, k, v = split_qkv(qkv.float(), nh, nkv)
    rep = nh // nkv
    mask = torch.from_numpy(m)
    exact = ref_attention(q, k.repeat_interleave(rep, 1), v.repeat_interleave(rep, 1), mask, scale)
    # the same attention on the dequantised operands (Q blocks along d, K blocks along d, V blocks of 32 keys along the tile)
    qmul = scale * 1.4426950408889634
    qd = torch.from_numpy(quantise_blocks((q.numpy().astype(np.float32) * np.float32(qmul)).astype(np.float64).reshape(B, nh, L, 3, 32))[0]).reshape(B, nh, L, D) / qmul
    kd = torch.from_numpy(quantise_blocks(k.numpy().astype(np.float64).reshape(B, nkv, L, 3, 32))[0]).reshape(B, nkv, L, D)
    Lp = (L + 31) // 32 * 32
    vpad = np.zeros((B, nkv, Lp, D)); vpad[:, :, :L] = v.numpy()
    vd = quantise_blocks(vpad.reshape(B, nkv, Lp // 32, 32, D).transpose(0, 1, 2, 4, 3))[0].transpose(0, 1, 2, 4, 3).reshape(B, nkv, Lp, D)[:, :, :L]
    deq = ref_attention(qd, kd.repeat_interleave(rep, 1), torch.from_numpy(vd).repeat_interleave(rep, 1), mask, scale)
    to_rows = lambda t: t.transpose(1, 2).reshape(B, L, nh * D)
    exact, deq = to_rows(exact), to_rows(deq)
    rows = [(b, 0, L) for b in range(B)] if segs is None else list(segs)
    for b, r0, r1 in rows:
        got = out[b, r0 - q_start:r1 - q_start]
        e_deq, e_exact = rel_l2(got, deq[b, r0:r1]), rel_l2(got, exact[b, r0:r1])
        print(f"fp8 attention rows [{r0},{r1}): rel-L2 vs dequantised-operand reference {e_deq:.3e}, vs exact {e_exact:.3e}")
        assert e_deq < 3e-2
        assert e_exact < 8e-2
    if segs is not None:   # rows outside the segments are not touched
        seen = np.zeros(L - q_start, dtype=bool)
        for _, r0, r1 in segs:
            seen[r0 - q_start:r1 - q_start] = True
        assert bool((out[0].cpu()[~torch.from_numpy(seen)].float() == 7.0).all())


def test_attention_fp8_late_spike_and_masked_first_tile(ops):
    """Running-maximum moves in the middle of the tile loop; first tile mixed for every row and wholly masked for some."""
    B, L, nh = 1, 900, 2
    q = torch.randn(B, nh, L, D, generator=g(44)).to(BF).float()
    k = torch.randn(B, nh, L, D, generator=g(45)).to(BF).float()
    v = torch.randn(B, nh, L, D, generator=g(46)).to(BF).float()
    k[:, :, 200] = (q[:, :, 300] * 8.0).to(BF).float()
    k[:, :, 700] = (q[:, :, 610] * 3.0).to(BF).float()
    m = np.ones((B, L, L), dtype=np.uint8)
    m[:, :, :17] = 0
    m[:, 500:, 17:64] = 0
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    qkv = torch.cat([t.transpose(1, 2).reshape(B, L, nh * D) for t in (q, k, v)], dim=-1).to(DEV, BF)
    out = ops.attention_qkv_fp8(qkv, pm, nh, nh, D)
    ref = ref_attention(q, k, v, torch.from_numpy(m), 1 / math.sqrt(D)).transpose(1, 2).reshape(B, L, -1)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 8e-2


def test_quantiser_row_begin_keeps_earlier_rows(ops):
    """vgpt_attn_fp8_quantize(row_begin): rows below row_begin keep the bytes the workspace holds (the engine's cached
    prefix), rows from it on are rewritten; same result as one full pass."""
    B, L, nh, nkv = 1, 300, 2, 2
    qkv = torch.randn(B, L, (nh + 2 * nkv) * D, generator=g(31)).to(BF).to(DEV)
    full = ops.attention_fp8_workspace(B, L, nh, nkv, D, DEV).zero_()
    ops.attention_fp8_quantize(qkv, full, nh, nkv, D)
    part = full.clone()
    qkv2 = qkv.clone()
    qkv2[:, 128:] = torch.randn(B, L - 128, qkv.shape[-1], generator=g(32)).to(BF).to(DEV)   # rows >= 128 change
    ops.attention_fp8_quantize(qkv2, part, nh, nkv, D, row_begin=128)
    want = ops.attention_fp8_workspace(B, L, nh, nkv, D, DEV).zero_()
    ops.attention_fp8_quantize(qkv2, want, nh, nkv, D)
    assert torch.equal(part, want) and not torch.equal(part, full)
    with pytest.raises(Exception, match="multiple of 64"):
        ops.attention_fp8_quantize(qkv, full, nh, nkv, D, row_begin=100)


def test_attention_fp8_rejects_other_head_dims(ops):
    pm = ops.pack_mask(torch.ones(1, 64, 64, dtype=torch.bool, device=DEV))
    with pytest.raises(Exception, match="96"):
        ops.attention_qkv_fp8(torch.zeros(1, 64, 3 * 2 * 128, dtype=BF, device=DEV), pm, 2, 2, 128)


@pytest.mark.parametrize("mode", ["hoist", "none"])
def test_sampler_with_fp8_attention(mode):
    """The sampler's fast path with attention_precision = "fp8" (engine.StaticDenoiser; the per-clip passes stay bf16):
    3 Euler steps with CFG on the tiny model (2 layers, 2 heads x 96) against the fp32 oracle and against the bf16
    engine.  Tolerance on the sampled latents: rel-L2 <= 6e-2 (bf16 path: 3e-2, tests/test_model_gpu.py)."""
    from oracle import restate as R
    from tests import smoke_case as SC
    S = importlib.import_module("video-gpt_amd.scheduler")
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    cfg, steps, C, G, hw = R.TINY, 3, 2, 2, (16, 16)
    bl = (hw[0] // 2) * (hw[1] // 2) + 2
    p, batch, z, cond = SC.build_case(cfg, C=C, G=G, hw=hw, use_cfg=True)
    lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)], (C + G) * bl)
    model = SC.build_product_model(cfg, p, DEV)
    outs = {}
    for prec in ("fp8", "bf16"):
        sched = S.LVMScheduler(num_steps=steps)
        sched.reuse_condition_prefix = sched.hoist_special_rows = mode == "hoist"
        sched.attention_precision = prec
        kw = SC.model_kwargs(batch, cond, DEV, use_cfg=True)
        kw["attention_mask"] = lay
        outs[prec] = torch.cat(sched([x.to(DEV, BF) for x in z], model.frame_block_forward_with_cfg, kw, prediction_type="x1"))
        assert sched.last_engine.attn_fp8 == (prec == "fp8")
    ref = torch.cat(SC.oracle_sample(cfg, p, batch, z, cond, steps, "x1", use_cfg=True))
    e_ref, e_bf = rel_l2(outs["fp8"], ref), rel_l2(outs["fp8"], outs["bf16"])
    print(f"sampler with fp8 attention ({mode}): rel-L2 vs oracle {e_ref:.3e}, vs bf16 engine {e_bf:.3e}; bf16 engine vs oracle {rel_l2(outs['bf16'], ref):.3e}")
    assert not torch.equal(outs["fp8"], outs["bf16"])
    assert e_ref < 6e-2 and e_bf < 6e-2


def test_engine_rejects_unknown_attention_precision():
    E = importlib.import_module("video-gpt_amd.engine")
    with pytest.raises(Exception, match="attention_precision"):
        E.StaticDenoiser(_NotReady(), None, None, None, None, None, None, None, 1, (2, 2), False, 1.0, attention_precision="int4")


class _NotReady:
    def _check_ready(self):
        return None


def test_attention_fp8_on_the_real_cfg2_mask(ops):
    """The sampler's own layout at cfg-2 (256^2, C = 4 condition + G = 8 generated frames, CFG row packed behind, special
    rows hoisted: L = 5248, 4096 live rows cut at the sequence seam) with 2 heads: fp8 attention of the live rows against
    fp64 on the unquantised inputs for row bands around the seams, and against the bf16 kernel on every row."""
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    nh, C, G, bl, N, nf = 2, 4, 8, 258, 256, 16
    lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)], (C + G) * bl)
    packed, _ = lay.pack()
    S0 = C * bl
    x_old = [S0 + f * bl + 2 for f in range(G)] + [S0 + G * bl + f * bl + 2 for f in range(G)]
    d_old = [x - 2 for x in x_old]; t_old = [x - 1 for x in x_old]
    S = (S0 + 2 * nf + 127) // 128 * 128
    perm = list(range(S0)) + d_old + t_old + [-1] * (S - S0 - 2 * nf) + [x + j for x in x_old for j in range(N)]
    lp = packed.permute(np.array(perm))
    pm = lp.packed_mask(DEV)
    L = len(perm)
    assert (L, S) == (5248, 1152)
    qkv = torch.randn(1, L, 3 * nh * D, generator=g(21)).to(BF)
    segs = ((0, S, S + 2048), (0, S + 2048, L))
    out8 = torch.empty(1, L - S, nh * D, dtype=BF, device=DEV)
    out16 = torch.empty_like(out8)
    ops.attention_qkv_fp8(qkv.to(DEV), pm, nh, nh, D, out=out8, q_start=S, segments=segs)
    ops.attention_qkv_range(qkv.to(DEV), pm, nh, nh, D, S, out16, segments=segs)
    assert torch.isfinite(out8).all()
    e16 = rel_l2(out8, out16)
    dense = torch.from_numpy(lp.to_bool())                          # (1, L, L)
    q, k, v = split_qkv(qkv.float(), nh, nh)
    worst = 0.0
    for r0 in (S, S + 1000, S + 2048 - 64, S + 2048, L - 128):      # first rows, mid, both sides of the seam, last rows
        rows = slice(r0, r0 + 64)
        ref = ref_attention(q[:, :, rows], k, v, dense[:, rows], 1 / math.sqrt(D)).transpose(1, 2).reshape(1, 64, nh * D)
        worst = max(worst, rel_l2(out8[:, r0 - S:r0 - S + 64], ref))
        assert rel_l2(out16[:, r0 - S:r0 - S + 64], ref) < 1e-2
    print(f"fp8 attention on the cfg-2 layout: rel-L2 vs bf16 kernel {e16:.3e}, worst band vs fp64 {worst:.3e}")
    assert e16 < 8e-2 and worst < 8e-2
