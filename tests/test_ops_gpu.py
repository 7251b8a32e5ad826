"""GPU parity of every HIP entry point against the CPU oracle (oracle/restate.py), called through
the C ABI (video-gpt_amd/ops.py -> libvgpt_hip.so).

Tolerances (bf16 kernels vs the fp32 oracle evaluated on the SAME bf16-rounded inputs):
  rel-L2 <= 4e-3 for single ops whose only error is the final bf16 rounding (bf16 eps = 7.8e-3,
  RMS rounding error ~ eps/ (2*sqrt(3)) = 2.3e-3), <= 1e-2 where an intermediate is rounded to
  bf16 as the reference's bf16 path also does (attention probabilities, RoPE tables).
Integer / bit work (mask packing, tile summary, gathers) is compared bit-exactly.
"""
import importlib
import math
import os

import numpy as np
import pytest
import torch

from oracle import restate as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
BF = torch.bfloat16


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def bf(x):  # round to bf16, keep fp32 container (oracle input)
    return x.to(BF).float()


def g(seed):
    return torch.Generator("cpu").manual_seed(seed)


# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("rows,H", [(5, 192), (37, 3072), (3, 4096), (2, 8192), (0, 64), (130, 1032)])
def test_rmsnorm(ops, rows, H):
    x = bf(torch.randn(rows, H, generator=g(1)) * 3)
    w = bf(1 + 0.1 * torch.randn(H, generator=g(2)))
    y = ops.rmsnorm(x.to(DEV, BF), w.to(DEV, BF), 1e-5)
    ref = R.rmsnorm(x, w, 1e-5)
    assert y.shape == (rows, H)
    if rows:
        assert rel_l2(y, ref) < 4e-3


def test_rmsnorm_rejects_odd_width(ops):
    with pytest.raises(Exception):
        ops.rmsnorm(torch.zeros(2, 100, device=DEV, dtype=BF), torch.ones(100, device=DEV, dtype=BF), 1e-5)


@pytest.mark.parametrize("hd,nh,nkv", [(96, 2, 2), (96, 32, 32), (128, 4, 2), (64, 3, 1)])
def test_rope(ops, hd, nh, nkv):
    B, L = 2, 77
    pos = torch.randint(0, 3100, (B, L), generator=g(3))
    qkv = bf(torch.randn(B, L, (nh + 2 * nkv) * hd, generator=g(4)))
    inv = ops.rope_inv_freq(hd, 10000.0, DEV)
    cos, sin = ops.rope_table(pos.to(DEV), inv, round_bf16=True)
    rc, rs = R.rope_cos_sin(pos, hd, 10000.0, BF)
    # table: bf16-rounded cos/sin of the fp32 angle (device sinf/cosf vs torch: allow 1 bf16 ulp)
    assert (cos.cpu() - rc[..., : hd // 2].float().reshape(-1, hd // 2)).abs().max() <= 2 ** -7
    assert (sin.cpu() - rs[..., : hd // 2].float().reshape(-1, hd // 2)).abs().max() <= 2 ** -7
    out = ops.rope_qk_inplace(qkv.to(DEV, BF).clone(), cos, sin, nh, nkv, hd).cpu().float()
    q = qkv[..., : nh * hd].view(B, L, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd:(nh + nkv) * hd].view(B, L, nkv, hd).transpose(1, 2)
    rq, rk = R.apply_rope(q, k, rc.float(), rs.float())
    assert rel_l2(out[..., : nh * hd], rq.transpose(1, 2).reshape(B, L, -1)) < 6e-3
    assert rel_l2(out[..., nh * hd:(nh + nkv) * hd], rk.transpose(1, 2).reshape(B, L, -1)) < 6e-3
    assert torch.equal(out[..., (nh + nkv) * hd:], qkv[..., (nh + nkv) * hd:])  # v untouched


@pytest.mark.parametrize("M,K,hd,nh,nkv", [(77, 192, 96, 2, 2), (300, 128, 64, 3, 1), (515, 64, 128, 4, 2),
                                            # 128-tile kernel with m tail; 256x192 / 256x256 tiles (qkv of cfg-2: 4096 x 9216)
                                            # (N = 9216 at 4096 / 2048 rows: the 256x288-tile kernel, two / one full rounds)
                                            (4096, 192, 96, 32, 32), (4100, 64, 96, 32, 32), (5160, 128, 96, 32, 32),
                                            (2048, 256, 96, 32, 32), (4100, 320, 96, 32, 32), (4096, 128, 64, 32, 16)])
def test_qkv_gemm_with_fused_rope(ops, M, K, hd, nh, nkv, gemm_family):
    if gemm_family == 1 and M < 2048:
        pytest.skip("small grids take the same kernel in both families")
    """vgpt_gemm_bf16_rope (qkv_proj + apply_rotary_pos_emb, sdpa_transform.py:39,52-53) vs the oracle's Linear (rounded
    to bf16) + apply_rope, and vs the two-kernel path (GEMM, then vgpt_rope_qk_inplace) it replaces."""
    N = (nh + 2 * nkv) * hd
    x = bf(torch.randn(M, K, generator=g(31)))
    w = bf(torch.randn(N, K, generator=g(32)) * 0.1)
    pos = torch.randint(0, 3100, (1, M), generator=g(33))
    cos, sin = ops.rope_table(pos.to(DEV), ops.rope_inv_freq(hd, 10000.0, DEV))
    fused = ops.linear_qkv_rope(x.to(DEV, BF), w.to(DEV, BF), cos, sin, nh, nkv, hd)
    two = ops.rope_qk_inplace(ops.linear(x.to(DEV, BF), w.to(DEV, BF)), cos, sin, nh, nkv, hd)
    # same arithmetic on the same bf16-rounded product; the compilers may contract a*c - b*s differently: <= 1 bf16 ulp
    d = (fused.float() - two.float()).abs()
    assert float((d / (two.float().abs() + 1e-3)).max()) <= 2 ** -7
    assert rel_l2(fused, two) < 1e-3
    assert torch.equal(fused[:, (nh + nkv) * hd:], two[:, (nh + nkv) * hd:])          # v columns: plain GEMM output
    qkv = bf(x.double() @ w.double().t()).float()[None]
    rc, rs = R.rope_cos_sin(pos, hd, 10000.0, BF)
    q = qkv[..., : nh * hd].view(1, M, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd:(nh + nkv) * hd].view(1, M, nkv, hd).transpose(1, 2)
    rq, rk = R.apply_rope(q, k, rc.float(), rs.float())
    out = fused.cpu().float()[None]
    assert rel_l2(out[..., : nh * hd], rq.transpose(1, 2).reshape(1, M, -1)) < 6e-3
    assert rel_l2(out[..., nh * hd:(nh + nkv) * hd], rk.transpose(1, 2).reshape(1, M, -1)) < 6e-3
    # empty input and shape errors
    assert ops.linear_qkv_rope(torch.empty(0, K, device=DEV, dtype=BF), w.to(DEV, BF), cos[:0], sin[:0], nh, nkv, hd).shape == (0, N)
    with pytest.raises(Exception):
        ops.linear_qkv_rope(x.to(DEV, BF), w.to(DEV, BF)[:-8], cos, sin, nh, nkv, hd)


@pytest.mark.parametrize("M,N,K", [(300, 576, 192), (128, 128, 64), (1, 4, 64), (258, 3072, 3072),
                                   (1000, 192, 512), (6192, 3072, 1024),
                                   # large grids take the 256x256-tile kernel (m tail / n tail)
                                   (5160, 9216, 192), (4000, 2052, 64),
                                   # grids the launch plan gives to the 256x192-tile kernel (exact fit / m and n tails)
                                   # ... and to the 256x288-tile kernel (N = 9216 = 32 tiles: 4096 / 2048 rows are whole rounds)
                                   (4096, 3072, 192), (4000, 3000, 128), (4096, 9216, 64), (2048, 9216, 320)])
@pytest.mark.parametrize("epi", ["none", "resid", "bias"])
def test_gemm(ops, M, N, K, epi, gemm_family):
    if gemm_family == 1 and M * N < 128 * 256 * 256:
        pytest.skip("small grids take the same kernel in both families")
    if epi != "none" and M > 1000 and N != 3000:
        pytest.skip("covered by the 'none' case")
    a = bf(torch.randn(M, K, generator=g(5)))
    w = bf(torch.randn(N, K, generator=g(6)) * 0.05)
    ref = a.double() @ w.double().t()
    kw = {}
    if epi == "resid":
        r = bf(torch.randn(M, N, generator=g(7)))
        kw["residual"] = r.to(DEV, BF)
        ref = ref + r.double()
    elif epi == "bias":
        b = bf(torch.randn(N, generator=g(8)))
        kw["bias"] = b.to(DEV, BF)
        ref = ref + b.double()
    y = ops.linear(a.to(DEV, BF), w.to(DEV, BF), **kw)
    assert y.shape == (M, N)
    assert rel_l2(y, ref) < 4e-3
    # element-wise: bf16 rounding of the exact value, 1 ulp slack for the fp32 accumulation order
    err = (y.cpu().double() - ref).abs()
    assert float((err / (ref.abs() + 1e-2)).max()) < 2e-2


def test_gemm_is_not_transposed(ops):
    """A = I with an asymmetric W catches swapped row/col maps (cdna guide §3)."""
    K = 128
    a = torch.eye(K)
    w = torch.arange(192 * K, dtype=torch.float32).reshape(192, K) % 251 - 125.0  # exact in bf16
    y = ops.linear(a.to(DEV, BF), w.to(DEV, BF))
    assert torch.equal(y.cpu().float(), w.t().contiguous())


def test_gemm_rejects_bad_k(ops):
    with pytest.raises(Exception):
        ops.linear(torch.zeros(4, 100, device=DEV, dtype=BF), torch.zeros(8, 100, device=DEV, dtype=BF))


@pytest.mark.parametrize("M,I,K,act", [(300, 512, 192, "silu"), (70, 64, 64, "gelu_pytorch_tanh"),
                                       (129, 8192, 3072, "silu"), (33, 144, 128, "gelu"),
                                       (2100, 8192, 256, "silu"), (5160, 4112, 64, "silu")])
def test_gated_mlp(ops, M, I, K, act, gemm_family):
    if gemm_family == 1 and M * I < 64 * 256 * 256:
        pytest.skip("small grids take the same kernel in both families")
    x = bf(torch.randn(M, K, generator=g(9)))
    w = bf(torch.randn(2 * I, K, generator=g(10)) * 0.05)
    gate, up = (x.double() @ w.double().t()).chunk(2, dim=-1)
    ref = up * R._ACT[act](gate)
    y = ops.gated_mlp_act(x.to(DEV, BF), w.to(DEV, BF), ops.act_code(act))
    assert rel_l2(y, ref) < 4e-3


def test_gemm_families_agree_bit_for_bit(ops):
    """The four-wave kernel (hand-scheduled loop, include/vgpt.h vgpt_gemm_set_family) adds a k-tile's products to an
    accumulator in the same order as the eight-wave kernels, so the two families are the SAME function bit for bit: plain +
    residual at the o_proj / down_proj shapes (192-wide tiles), the gated activation incl. the stored [gate | up] (256-wide),
    ragged rows and columns."""
    import importlib
    lib = importlib.import_module("video-gpt_amd._lib").load()
    def both(fn):
        out = []
        for fam in (0, 1):
            prev = lib.vgpt_gemm_set_family(fam)
            try:
                out.append(fn())
            finally:
                lib.vgpt_gemm_set_family(prev)
        return out
    for (M, N, K) in ((4096, 3072, 3072), (4096, 3072, 8192), (4100, 3076, 320), (4096, 9216, 384)):   # the last: 256 x 288 tiles
        a = bf(torch.randn(M, K, generator=g(71))).to(DEV, BF)
        w = bf(torch.randn(N, K, generator=g(72)) * 0.05).to(DEV, BF)
        r = bf(torch.randn(M, N, generator=g(73))).to(DEV, BF)
        y4, y8 = both(lambda: ops.linear(a, w, residual=r))
        assert torch.equal(y4, y8), (M, N, K)
    M, I, K = 4100, 4112, 256
    x = bf(torch.randn(M, K, generator=g(74))).to(DEV, BF)
    w = bf(torch.randn(2 * I, K, generator=g(75)) * 0.05).to(DEV, BF)
    def keep():
        gu = torch.full((M, 2 * I), 7.0, dtype=BF, device=DEV)
        return ops.gated_mlp_act(x, w, ops.ACT_SILU, gate_up_out=gu), gu
    (a4, g4), (a8, g8) = both(keep)
    assert torch.equal(g4, g8) and torch.equal(a4, a8)
    p4, p8 = both(lambda: ops.gated_mlp_act(x, w, ops.ACT_SILU))
    assert torch.equal(p4, p8)


# ---- RMSNorm folded into the GEMMs around it (include/vgpt.h: vgpt_gemm_bf16_resid_ssq -> *_prenorm) ----------------------

def _hf_rmsnorm(x, w, eps):
    """Phi3RMSNorm.forward with its two bf16 roundings, on fp32 copies of bf16 values."""
    rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return bf(bf(x * rstd) * w)


def test_fold_norm_gain_and_row_statistics(ops):
    g_ = g(81)
    W = bf(torch.randn(520, 384, generator=g_) * 0.1).to(DEV, BF)
    gain = bf(1.0 + 0.2 * torch.randn(384, generator=g_)).to(DEV, BF)
    folded = ops.fold_norm_gain(W, gain)
    assert torch.equal(folded, (W.float() * gain.float()[None, :]).to(BF))
    x = bf(torch.randn(777, 384, generator=g_)).to(DEV, BF)
    rs = ops.rms_rstd(x, 1e-5)
    assert rs.shape == (777,)
    assert torch.allclose(rs.cpu().double(), torch.rsqrt(x.cpu().double().pow(2).mean(-1) + 1e-5), rtol=1e-5)


def test_linear_resid_rstd_is_the_residual_gemm_plus_its_rows_rstd(ops):
    """The producer side: same output bit for bit as linear(..., residual=), 1 / rms of the ROUNDED output rows (the partial
    sums of the tile columns added up by the last workgroup of every 256-row block), in place on the residual stream, ragged
    M and N (columns past N excluded), the workspace's counters back at zero after every launch, deterministic."""
    eps = 1e-5
    for (M, N, K) in ((4096, 2048, 256), (4000, 3076, 128)):
        nbytes = ops.norm_workspace_bytes(M, N, K)
        assert nbytes > 0
        ws = ops.norm_workspace(nbytes, DEV)
        a = bf(torch.randn(M, K, generator=g(82))).to(DEV, BF)
        w = bf(torch.randn(N, K, generator=g(83)) * 0.1).to(DEV, BF)
        r = bf(torch.randn(M, N, generator=g(84))).to(DEV, BF)
        ref = ops.linear(a, w, residual=r)
        want = torch.rsqrt(ref.cpu().double().pow(2).mean(-1) + eps)
        rstd = torch.full((M,), float("nan"), dtype=torch.float32, device=DEV)
        hid = r.clone()
        out = ops.linear_resid_rstd(a, w, hid, rstd, ws, eps, out=hid)          # in place
        assert out.data_ptr() == hid.data_ptr() and torch.equal(hid, ref)
        assert torch.allclose(rstd.cpu().double(), want, rtol=3e-6)
        n_cnt = -(-M // 256)
        assert int(ws[: n_cnt * 4].view(torch.int32).abs().sum()) == 0          # every counter is back at zero
        for _ in range(3):                                                      # the same workspace again, fresh outputs
            again = torch.full((M,), float("nan"), dtype=torch.float32, device=DEV)
            ops.linear_resid_rstd(a, w, r, again, ws, eps, out=torch.empty_like(r))
            assert torch.equal(again, rstd)                                     # deterministic (no float atomics)
    assert ops.norm_workspace_bytes(300, 512, 256) == 0                         # small grids keep the separate norm
    with pytest.raises(Exception):
        ops.linear_resid_rstd(a, w, r, rstd, ws[:256], eps, out=torch.empty_like(r))


def test_prenorm_consumers_equal_norm_then_gemm(ops, gemm_family):
    """The consumer side on both kernel families: qkv_proj + RoPE and gate_up + activation on the RAW stream with the gain
    folded into the weight and the rows scaled by 1 / rms, against (a) the reference sequence in fp64 on Phi3RMSNorm's own
    rounded output, (b) the unfused kernels (rmsnorm, then GEMM) -- the two differ by where the bf16 roundings sit, well inside
    the single-op tolerance."""
    M, H, hd, nq, nkv, I, eps = 4096, 512, 96, 8, 8, 2048, 1e-5
    g_ = g(85)
    x = bf(torch.randn(M, H, generator=g_) * torch.rand(M, 1, generator=g_).mul(3).add(0.2))     # rows of very different norms
    gain = bf(1.0 + 0.3 * torch.randn(H, generator=g_))
    wq = bf(torch.randn((nq + 2 * nkv) * hd, H, generator=g_) * 0.05)
    wgu = bf(torch.randn(2 * I, H, generator=g_) * 0.05)
    xd, gd, wqd, wgud = x.to(DEV, BF), gain.to(DEV, BF), wq.to(DEV, BF), wgu.to(DEV, BF)
    rstd = ops.rms_rstd(xd, eps)
    pos = torch.randint(0, 3100, (1, M), generator=g_)
    cos, sin = ops.rope_table(pos.to(DEV), ops.rope_inv_freq(hd, 10000.0, DEV))
    nrm = ops.rmsnorm(xd, gd, eps)
    assert rel_l2(nrm, _hf_rmsnorm(x, gain, eps)) < 4e-3
    # qkv + RoPE
    fused = ops.linear_qkv_rope_prenorm(xd, ops.fold_norm_gain(wqd, gd), cos, sin, rstd, nq, nkv, hd,
                                        out=torch.empty(M, wq.shape[0], dtype=BF, device=DEV))
    two = ops.linear_qkv_rope(nrm, wqd, cos, sin, nq, nkv, hd)
    assert rel_l2(fused, two.float().cpu()) < 4e-3
    nrm64 = _hf_rmsnorm(x, gain, eps).double()
    qkv = bf((nrm64 @ wq.double().t()).float()).float()[None]
    rc, rs = R.rope_cos_sin(pos, hd, 10000.0, BF)
    q = qkv[..., : nq * hd].view(1, M, nq, hd).transpose(1, 2)
    k = qkv[..., nq * hd:(nq + nkv) * hd].view(1, M, nkv, hd).transpose(1, 2)
    rq, rk = R.apply_rope(q, k, rc.float(), rs.float())
    out = fused.cpu().float()[None]
    assert rel_l2(out[..., : nq * hd], rq.transpose(1, 2).reshape(1, M, -1)) < 6e-3
    assert rel_l2(out[..., nq * hd:(nq + nkv) * hd], rk.transpose(1, 2).reshape(1, M, -1)) < 6e-3
    assert rel_l2(out[..., (nq + nkv) * hd:], qkv[..., (nq + nkv) * hd:]) < 4e-3
    # gate_up + activation
    fused = ops.gated_mlp_act_prenorm(xd, ops.fold_norm_gain(wgud, gd), rstd, ops.ACT_SILU, out=torch.empty(M, I, dtype=BF, device=DEV))
    two = ops.gated_mlp_act(nrm, wgud, ops.ACT_SILU)
    gate, up = (nrm64 @ wgu.double().t()).chunk(2, dim=-1)
    ref = up * R._ACT["silu"](gate)
    assert rel_l2(fused, ref) < 5e-3 and rel_l2(two, ref) < 5e-3 and rel_l2(fused, two.float().cpu()) < 5e-3


# ---------------------------------------------------------------------------------------------

def _np_bits(mask: np.ndarray) -> np.ndarray:
    B, L, _ = mask.shape
    W = (L + 31) // 32
    padded = np.zeros((B, L, W * 32), dtype=np.uint8)
    padded[:, :, :L] = mask
    return np.packbits(padded.reshape(B, L, W, 32), axis=-1, bitorder="little").view("<u4").reshape(B, L, W)


def _np_summary(mask: np.ndarray) -> np.ndarray:
    B, L, _ = mask.shape
    nqb, nkt = (L + 127) // 128, (L + 63) // 64
    out = np.zeros((B, nqb, nkt), dtype=np.uint8)
    for b in range(B):
        for qb in range(nqb):
            for kt in range(nkt):
                byte = 0
                for sub in range(4):
                    q0 = qb * 128 + sub * 32
                    rows = mask[b, q0:min(q0 + 32, L), kt * 64:min(kt * 64 + 64, L)]
                    if rows.shape[0] == 0 or not rows.any():
                        code = 0
                    elif rows.all() and kt * 64 + 64 <= L:
                        code = 1
                    else:
                        code = 2
                    byte |= code << (2 * sub)
                out[b, qb, kt] = byte
    return out


def _random_block_mask(B, L, seed):
    rng = np.random.default_rng(seed)
    m = np.zeros((B, L, L), dtype=np.uint8)
    for b in range(B):
        cuts = np.sort(rng.choice(np.arange(1, L), size=min(5, L - 1), replace=False))
        bounds = [0, *cuts.tolist(), L]
        for i in range(len(bounds) - 1):
            for j in range(i + 1):
                if rng.random() < 0.7 or i == j:
                    m[b, bounds[i]:bounds[i + 1], bounds[j]:bounds[j + 1]] = 1
        m[b] |= np.eye(L, dtype=np.uint8)  # no empty rows
        flip = rng.random((L, L)) < 0.02
        m[b] &= ~(flip & ~np.eye(L, dtype=bool))
    return m


@pytest.mark.parametrize("B,L", [(2, 72), (1, 300), (2, 129), (1, 64), (1, 33)])
def test_mask_pack_and_summary(ops, B, L):
    m = _random_block_mask(B, L, 11)
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV).bool())
    assert np.array_equal(pm.bits.cpu().numpy().view(np.uint32), _np_bits(m))
    assert np.array_equal(pm.summary.cpu().numpy(), _np_summary(m))
    assert pm.count_empty_rows() == 0
    # additive form of the same mask (OmniGen/transformer.py:139-145)
    for dt in (BF, torch.float32):
        add = R.additive_mask(torch.from_numpy(m).bool(), dt)
        pm2 = ops.pack_mask(add.to(DEV))
        assert torch.equal(pm2.bits, pm.bits) and torch.equal(pm2.summary, pm.summary)


def test_mask_collator_layout(ops):
    batch = R.collate_inference(4, 8, 256)
    m = batch["attention_mask"].numpy().astype(np.uint8)
    pm = ops.pack_mask(batch["attention_mask"].to(DEV))
    assert np.array_equal(pm.bits.cpu().numpy().view(np.uint32), _np_bits(m))
    assert np.array_equal(pm.summary.cpu().numpy(), _np_summary(m))


def test_mask_empty_rows_counted(ops):
    m = np.ones((1, 40, 40), dtype=np.uint8)
    m[0, 7] = 0
    m[0, 39] = 0
    assert ops.pack_mask(torch.from_numpy(m).to(DEV)).count_empty_rows() == 2


def _ref_attention(q, k, v, mask, scale):
    """(B,h,L,d) fp64 reference with the reference's additive-min mask semantics."""
    s = torch.matmul(q.double(), k.double().transpose(2, 3)) * scale
    s = s.masked_fill(~mask[:, None].bool(), float("-inf"))
    return torch.matmul(torch.softmax(s, dim=-1), v.double())


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,L,nh,nkv,hd", [(2, 72, 2, 2, 96), (1, 300, 3, 3, 96), (2, 129, 4, 2, 128),
                                           (1, 200, 2, 1, 64), (1, 1, 1, 1, 96)])
def test_attention_fused_qkv(ops, variant, B, L, nh, nkv, hd):
    m = _random_block_mask(B, L, 12)
    qkv = bf(torch.randn(B, L, (nh + 2 * nkv) * hd, generator=g(13)))
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    out = ops.attention_qkv(qkv.to(DEV, BF), pm, nh, nkv, hd, variant=variant)
    q = qkv[..., : nh * hd].view(B, L, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd:(nh + nkv) * hd].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    v = qkv[..., (nh + nkv) * hd:].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    ref = _ref_attention(q, k, v, torch.from_numpy(m), 1 / math.sqrt(hd)).transpose(1, 2).reshape(B, L, -1)
    assert rel_l2(out, ref) < 1e-2
    assert float((out.cpu().double() - ref).abs().max()) < 0.05


@pytest.mark.parametrize("B,L,q_start", [(1, 700, 0), (2, 1300, 0), (1, 1500, 256), (1, 64, 0)])
def test_qblock_order(ops, B, L, q_start):
    """Longest-first launch order == stable descending sort of the non-empty key tiles per q block."""
    m = _random_block_mask(B, L, 21)
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    got = pm.order(q_start).cpu().numpy()
    cnt = (_np_summary(m) != 0).sum(-1)[:, q_start // 128:]
    for b in range(B):
        want = np.argsort(-cnt[b], kind="stable") + q_start // 128
        assert np.array_equal(got[b], want)


@pytest.mark.parametrize("B,L,nh,nkv", [(1, 700, 8, 8), (2, 520, 4, 2), (1, 1100, 16, 4)])
def test_attention_launch_order_is_result_neutral(ops, B, L, nh, nkv):
    """The ordered, XCD-grouped work mapping (n_heads * B % 8 == 0) computes exactly what the plain mapping does."""
    import importlib
    lib = importlib.import_module("video-gpt_amd._lib")
    hd = 96
    m = _random_block_mask(B, L, 22)
    # make the q blocks unequal so the order is not the identity
    m[:, : L // 3, L // 2:] = 0
    for b in range(B):
        np.fill_diagonal(m[b], 1)
    qkv = bf(torch.randn(B, L, (nh + 2 * nkv) * hd, generator=g(23))).to(DEV, BF)
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    out = ops.attention_qkv(qkv, pm, nh, nkv, hd, variant=2)   # 4-wave kernel, ordered launch
    plain = torch.empty_like(out)
    width = qkv.shape[-1]
    kq = qkv.data_ptr() + nh * hd * 2
    vq = kq + nkv * hd * 2
    sb = L * width
    lib.call("vgpt_attn_blockmask_fwd", qkv.data_ptr(), kq, vq, plain.data_ptr(), pm.bits.data_ptr(), pm.summary.data_ptr(),
             B, L, nh, nkv, hd, sb, hd, width, sb, hd, width, sb, hd, width, L * nh * hd, hd, nh * hd,
             1.0 / math.sqrt(hd), 0, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(out, plain)
    assert not np.array_equal(pm.order(0).cpu().numpy()[0], np.arange((L + 127) // 128))


def _np_item_summary(m, items):
    B, L, _ = m.shape
    nkt = (L + 63) // 64
    out = np.zeros((len(items), nkt), dtype=np.uint16)
    for n, (b, r0, nr, _) in enumerate(items):
        for kt in range(nkt):
            code = 0
            for s_ in range(8):
                rows = m[b, r0 + 32 * s_: min(r0 + 32 * s_ + 32, r0 + nr), kt * 64: kt * 64 + 64]
                if rows.shape[0] == 0 or not rows.any():
                    c = 0
                elif rows.shape[1] == 64 and rows.all():
                    c = 1
                else:
                    c = 2
                code |= c << (2 * s_)
            out[n, kt] = code
    return out


@pytest.mark.parametrize("B,L,segs", [(1, 700, None), (2, 530, None), (1, 900, ((0, 0, 300), (0, 300, 316), (0, 316, 900))),
                                      (1, 64, None), (1, 1300, ((0, 256, 1300),))])
def test_attention_plan_build(ops, B, L, segs):
    """Items, their 16-bit tile summaries and the longest-first order (vgpt_attn_plan_build) against numpy."""
    item_rows = ops.ITEM_ROWS
    m = _random_block_mask(B, L, 31)
    m[:, : L // 3, L // 2:] = 0
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    plan = pm.plan(segs)
    items = plan.items.cpu().numpy()
    want_items = [(b, r, min(item_rows, r1 - r), 0) for b, r0, r1 in (segs or [(b, 0, L) for b in range(B)])
                  for r in range(r0, r1, item_rows)]
    assert items.tolist() == [list(t) for t in want_items]
    summ = plan.summary.cpu().numpy().view(np.uint16)[: len(want_items)]
    want = _np_item_summary(m, want_items)
    assert np.array_equal(summ, want)
    cnt = np.array([(want[i] != 0).sum() for i in range(len(want_items))])
    assert np.array_equal(plan.order.cpu().numpy()[: len(want_items)], np.argsort(-cnt, kind="stable"))
    # both tables live in one workspace sized by the C ABI's query (include/vgpt.h)
    nkt = (L + 63) // 64
    assert plan.workspace.numel() == ((len(want_items) * nkt * 2 + 255) // 256 + (len(want_items) * 4 + 255) // 256) * 256


@pytest.mark.parametrize("hd", [96, 128, 64])
@pytest.mark.parametrize("B,L,nh,nkv,segs", [(1, 700, 8, 8, None), (2, 530, 4, 2, None), (1, 1100, 16, 4, None),
                                             (1, 900, 3, 3, ((0, 0, 300), (0, 300, 316), (0, 316, 900))),
                                             (1, 1300, 8, 8, ((0, 256, 790), (0, 790, 1300)))])
def test_attention_planned_kernel(ops, B, L, nh, nkv, segs, hd):
    """Planned launches (arbitrary row segments cut into 128-row items, longest-first order) against the fp64 reference
    and, bit for bit, against the same kernel on aligned q blocks."""
    m = _random_block_mask(B, L, 32)
    m[:, : L // 3, L // 2:] = 0
    for b in range(B):
        np.fill_diagonal(m[b], 1)
    qkv = bf(torch.randn(B, L, (nh + 2 * nkv) * hd, generator=g(33)))
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    dq = qkv.to(DEV, BF)
    legacy = ops.attention_qkv(dq, pm, nh, nkv, hd, variant=2)
    if segs is None:
        out = ops.attention_qkv(dq, pm, nh, nkv, hd)
        covered = [(b, 0, L) for b in range(B)]
    else:
        out = torch.full((B, L, nh * hd), 7.0, dtype=BF, device=DEV)
        ops.attention_qkv_range(dq, pm, nh, nkv, hd, 0, out, segments=segs)
        covered = list(segs)
    q = qkv[..., : nh * hd].view(B, L, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd:(nh + nkv) * hd].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    v = qkv[..., (nh + nkv) * hd:].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    ref = _ref_attention(q, k, v, torch.from_numpy(m), 1 / math.sqrt(hd)).transpose(1, 2).reshape(B, L, -1)
    seen = torch.zeros(B, L, dtype=torch.bool)
    for b, r0, r1 in covered:
        seen[b, r0:r1] = True
        assert rel_l2(out[b, r0:r1], ref[b, r0:r1]) < 1e-2
        assert torch.equal(out[b, r0:r1], legacy[b, r0:r1])   # same kernel, same per-row operation sequence
    assert bool((out.cpu()[~seen].float() == 7.0).all())   # rows outside the segments are not touched
    if segs is not None and B == 1:
        # 256-row items = the eight-wave kernel (head dim 96; other head dims fall back to 128-row items): one K / V tile
        # staged per 256 query rows, every row still walks the same tiles in the same order -> the same bits
        out8 = torch.full((B, L, nh * hd), 7.0, dtype=BF, device=DEV)
        ops.attention_qkv_range(dq, pm, nh, nkv, hd, 0, out8, segments=segs, item_rows=256)
        assert torch.equal(out8, out)
        assert (pm.plan(segs, 256).item_rows == 256) and pm.plan(segs, 256).n_items < pm.plan(segs).n_items


@pytest.mark.parametrize("kind", ["dense", "packed", "frame_causal", "causal", "holes", "gqa_segments", "tiny_40", "tiny_64", "tiny_100",
                                  "tiny_130", "tiny_200"])
def test_attention_hand_scheduled_bodies_agree_bit_for_bit(ops, kind):
    """Head dim 96 runs its tiles through the hand-scheduled, software-pipelined bodies of csrc/gen/attn_p2_gen.py (the
    default); vgpt_attn_set_hand_scheduled(0) selects the compiler-scheduled tile body.  Same instructions per element and the same summation
    order: outputs AND log-sum-exp must agree bit for bit on dense, packed, block-causal, causal and holed masks (tiles a
    wave sees in full, in part, or not at all; wholly masked rows; a last tile running past L), with grouped KV heads, on
    row segments, and where a row's running maximum moves late (the rescale branch)."""
    ops_train = importlib.import_module("video-gpt_amd.ops_train")
    nh, nkv, hd = 4, 4, 96

    def frame_causal(L, f):
        i = np.arange(L) // f
        return (i[:, None] >= i[None, :]).astype(np.uint8)
    segs = None
    if kind == "dense":
        m = np.ones((1, 1100, 1100), dtype=np.uint8)
    elif kind == "packed":
        m = np.zeros((1, 1300, 1300), dtype=np.uint8); m[0, :780, :780] = 1; m[0, 780:, 780:] = 1
    elif kind == "frame_causal":
        m = np.stack([frame_causal(900, 300), frame_causal(900, 180)])
    elif kind == "causal":
        m = np.tril(np.ones((1, 700, 700), dtype=np.uint8))
    elif kind == "holes":
        m = frame_causal(1000, 100)[None].copy(); m[0, 400:440] = 0; m[0, :, 64:128] = 0; m[0, 5, 64] = 1
    elif kind.startswith("tiny"):        # one, two, three and four key tiles: prologue + drain only, then one / two pipelined bodies
        Lt = int(kind.split("_")[1])
        m = np.stack([np.ones((Lt, Lt), dtype=np.uint8), frame_causal(Lt, 16)])
    else:
        nh, nkv = 8, 2
        m = _random_block_mask(1, 1300, 32)
        np.fill_diagonal(m[0], 1)
        segs = ((0, 256, 790), (0, 790, 1300))
    B, L = m.shape[:2]
    qkv = bf(torch.randn(B, L, (nh + 2 * nkv) * hd, generator=g(71)) * 1.5)
    kq = qkv[..., nh * hd:(nh + nkv) * hd].view(B, L, nkv, hd)
    kq[:, (2 * L) // 3] = bf(qkv[..., : nh * hd].view(B, L, nh, hd)[:, L - 7, :nkv] * (6.0 if L > 256 else 2.0))   # a late key far above the others
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    dq = qkv.to(DEV, BF)
    res = {}
    lib = importlib.import_module("video-gpt_amd._lib").load()
    prev = lib.vgpt_attn_set_hand_scheduled(1)
    try:
        for p2 in ("0", "1"):
            lib.vgpt_attn_set_hand_scheduled(int(p2))
            if segs is None:
                out = torch.empty(B, L, nh * hd, dtype=BF, device=DEV)
                lse = torch.empty(B, nh, L, dtype=torch.float32, device=DEV)
                ops_train.attention_qkv_train(dq, pm, nh, nkv, hd, out, lse)
                res[p2] = (out, lse)
            else:
                out = torch.full((B, L, nh * hd), 7.0, dtype=BF, device=DEV)
                ops.attention_qkv_range(dq, pm, nh, nkv, hd, 0, out, segments=segs)
                res[p2] = (out, torch.zeros(1, device=DEV))
    finally:
        lib.vgpt_attn_set_hand_scheduled(prev)
    assert torch.equal(res["0"][0], res["1"][0])
    assert torch.equal(res["0"][1], res["1"][1])
    q = qkv[..., : nh * hd].view(B, L, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd:(nh + nkv) * hd].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    v = qkv[..., (nh + nkv) * hd:].view(B, L, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    ref = _ref_attention(q, k, v, torch.from_numpy(m), 1 / math.sqrt(hd)).transpose(1, 2).reshape(B, L, -1)
    live = torch.from_numpy(m.any(axis=2))                    # rows that see no key at all come out as zeros
    if segs is not None:
        live[:, :256] = False
    assert rel_l2(res["1"][0].cpu()[live], ref[live]) < 1e-2
    assert bool((res["1"][0].cpu()[~torch.from_numpy(m.any(axis=2))].float() == 0).all())


@pytest.mark.parametrize("spike_at,boost", [(200, 8.0), (40, 30.0), (700, 3.0)])
def test_attention_late_spike(ops, spike_at, boost):
    """One key far above the others late in the sequence: a row's running maximum must move in the middle of the tile
    loop (and the first tile is mixed for every row, wholly masked for some)."""
    B, L, nh, hd = 1, 900, 2, 96
    q = bf(torch.randn(B, nh, L, hd, generator=g(44)))
    k = bf(torch.randn(B, nh, L, hd, generator=g(45)))
    v = bf(torch.randn(B, nh, L, hd, generator=g(46)))
    k[:, :, spike_at] = bf(q[:, :, 300] * boost)
    k[:, :, spike_at + 130] = bf(q[:, :, 610] * boost)
    m = np.ones((B, L, L), dtype=np.uint8)
    m[:, :, :17] = 0
    m[:, 500:, 17:64] = 0
    pm = ops.pack_mask(torch.from_numpy(m).to(DEV))
    qkv = torch.cat([t.transpose(1, 2).reshape(B, L, nh * hd) for t in (q, k, v)], dim=-1).to(DEV, BF)
    out = ops.attention_qkv(qkv, pm, nh, nh, hd)
    ref = _ref_attention(q, k, v, torch.from_numpy(m), 1 / math.sqrt(hd)).transpose(1, 2).reshape(B, L, -1)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) < 1e-2
    assert float((out.cpu().float() - ref.float()).abs().max()) < 6e-2   # |O| up to ~4: a few bf16 ulps


@pytest.mark.parametrize("variant", [0, 1])
def test_attention_forced_rescale(ops, variant):
    """Spike one key late in the sequence so the running max jumps at a later tile (online-softmax rescale)."""
    B, L, nh, hd = 1, 256, 1, 96
    q = bf(torch.randn(B, nh, L, hd, generator=g(14)))
    k = bf(torch.randn(B, nh, L, hd, generator=g(15)))
    v = bf(torch.randn(B, nh, L, hd, generator=g(16)))
    k[0, 0, 200] = q[0, 0, 5] * 4.0  # huge score for (q=5, key=200), third tile
    mask = torch.ones(B, L, L, dtype=torch.bool)
    out = ops.sdpa(q.to(DEV, BF), k.to(DEV, BF), v.to(DEV, BF), attn_mask=mask.to(DEV), variant=variant)
    ref = _ref_attention(q, k, v, mask, 1 / math.sqrt(hd))
    assert rel_l2(out, ref) < 1e-2
    assert float((out.cpu().double()[0, 0, 5] - ref[0, 0, 5]).abs().max()) < 0.05


def test_attention_sdpa_seam_real_layout(ops):
    """The local_attn slot on (B,h,S,d) tensors with the 256^2 / C=4 / G=8 collator mask, additive form."""
    batch = R.collate_inference(4, 8, 256)
    mask = batch["attention_mask"]
    B, L = mask.shape[:2]
    nh, hd = 2, 96
    q = bf(torch.randn(B, nh, L, hd, generator=g(17)))
    k = bf(torch.randn(B, nh, L, hd, generator=g(18)))
    v = bf(torch.randn(B, nh, L, hd, generator=g(19)))
    add = R.additive_mask(mask, BF)
    out = ops.sdpa(q.to(DEV, BF), k.to(DEV, BF), v.to(DEV, BF), attn_mask=add.to(DEV), dropout_p=0.0,
                   is_causal=False)
    ref = _ref_attention(q, k, v, mask, 1 / math.sqrt(hd))
    assert rel_l2(out, ref) < 1e-2
    out1 = ops.sdpa(q.to(DEV, BF), k.to(DEV, BF), v.to(DEV, BF), attn_mask=mask.to(DEV), variant=1)
    assert rel_l2(out1, ref) < 1e-2


# ---------------------------------------------------------------------------------------------

def test_embed_gather(ops):
    table = bf(torch.randn(64, 192, generator=g(20)))
    ids = torch.randint(0, 64, (2, 37), generator=g(21))
    out = ops.embed_gather(ids.to(DEV), table.to(DEV, BF))
    assert torch.equal(out.cpu().float(), table[ids])


@pytest.mark.parametrize("cfg,hw", [(R.TINY, (8, 8)), (R.TINY, (4, 12)), (R.Phi3Cfg(), (32, 32))])
def test_patch_embed(ops, cfg, hw):
    h, w = hw
    H = cfg.hidden_size
    nf = 3
    x = bf(torch.randn(nf, 4, h, w, generator=g(22)))
    wt = bf(torch.randn(H, 4, 2, 2, generator=g(23)) * 0.2)
    b = bf(torch.randn(H, generator=g(24)) * 0.1)
    pos = bf(R.make_pos_embed(cfg))
    ntok = (h // 2) * (w // 2)
    rows = ntok * nf + 7
    seq = torch.full((rows, H), 9.0, dtype=BF, device=DEV)
    dst = torch.tensor([3, 3 + 2 * ntok, 3 + ntok], dtype=torch.int32, device=DEV)
    ops.patch_embed(x.to(DEV, BF), wt.to(DEV, BF), b.to(DEV, BF), pos[0].to(DEV, BF), dst, seq, cfg.pos_embed_max_size)
    seq = seq.cpu().float()
    for f, r0 in enumerate(dst.tolist()):
        ref = R.patch_embed(x[f:f + 1], wt, b, 2) + R.cropped_pos_embed(pos, cfg, h, w)
        assert rel_l2(seq[r0:r0 + ntok], ref[0]) < 4e-3
    assert torch.all(seq[:3] == 9.0) and torch.all(seq[3 + 3 * ntok:] == 9.0)


def test_timestep_sinusoid_and_mlp(ops):
    H = 192
    t = torch.tensor([0.0, 0.02, 0.37, 0.5, 0.9, 1.0])
    freqs = ops.timestep_freqs(256, DEV)
    emb = ops.timestep_sinusoid(t.to(DEV), freqs)
    ref = R.timestep_embedding(t, 256)
    assert (emb.cpu().float() - ref).abs().max() <= 2 ** -8
    p = {"tt.mlp.0.weight": bf(torch.randn(H, 256, generator=g(25)) * 0.05), "tt.mlp.0.bias": bf(torch.randn(H, generator=g(26)) * 0.05),
         "tt.mlp.2.weight": bf(torch.randn(H, H, generator=g(27)) * 0.05), "tt.mlp.2.bias": bf(torch.randn(H, generator=g(28)) * 0.05)}
    d = {k: v.to(DEV, BF) for k, v in p.items()}
    h1 = ops.linear_small(emb, d["tt.mlp.0.weight"], d["tt.mlp.0.bias"], post_act=ops.ACT_SILU)
    out = ops.linear_small(h1, d["tt.mlp.2.weight"], d["tt.mlp.2.bias"])
    ref = R.timestep_embedder(p, "tt", t, torch.float32)
    assert rel_l2(out, ref) < 1e-2


@pytest.mark.parametrize("M,N,K", [(1, 10, 64), (16, 3072, 3072), (32, 100, 256), (9, 6144, 192)])
def test_linear_small(ops, M, N, K):
    x = bf(torch.randn(M, K, generator=g(29)))
    w = bf(torch.randn(N, K, generator=g(30)) * 0.05)
    b = bf(torch.randn(N, generator=g(31)))
    out = ops.linear_small(x.to(DEV, BF), w.to(DEV, BF), b.to(DEV, BF), pre_act=ops.ACT_SILU)
    ref = bf(torch.nn.functional.silu(x)).double() @ w.double().t() + b.double()
    assert rel_l2(out, ref) < 4e-3
    # scattered rows (time tokens written into the sequence)
    rows = torch.randperm(M + 5, generator=g(32))[:M].to(torch.int32)
    seq = torch.zeros(M + 5, N, dtype=BF, device=DEV)
    ops.linear_small(x.to(DEV, BF), w.to(DEV, BF), b.to(DEV, BF), out=seq, out_row=rows.to(DEV), ldo=N)
    ref2 = x.double() @ w.double().t() + b.double()
    assert rel_l2(seq.cpu()[rows.long()], ref2) < 4e-3


@pytest.mark.parametrize("cfg,hw", [(R.TINY, (8, 8)), (R.TINY, (4, 12)), (R.Phi3Cfg(), (32, 32))])
def test_final_layer(ops, cfg, hw):
    h, w = hw
    H = cfg.hidden_size
    nf = 3
    ntok = (h // 2) * (w // 2)
    hidden = bf(torch.randn(nf * ntok + 5, H, generator=g(33)))
    c = bf(torch.randn(nf, H, generator=g(34)))
    p = {"final_layer.adaLN_modulation.1.weight": bf(torch.randn(2 * H, H, generator=g(35)) * 0.05),
         "final_layer.adaLN_modulation.1.bias": bf(torch.randn(2 * H, generator=g(36)) * 0.05),
         "final_layer.linear.weight": bf(torch.randn(16, H, generator=g(37)) * 0.05),
         "final_layer.linear.bias": bf(torch.randn(16, generator=g(38)) * 0.05)}
    d = {k: v.to(DEV, BF) for k, v in p.items()}
    mod = ops.linear_small(c.to(DEV, BF), d["final_layer.adaLN_modulation.1.weight"],
                           d["final_layer.adaLN_modulation.1.bias"], pre_act=ops.ACT_SILU)
    src = torch.tensor([2, 2 + 2 * ntok, 2 + ntok], dtype=torch.int32, device=DEV)
    out = torch.empty(nf, 4, h, w, dtype=BF, device=DEV)
    ops.final_layer(hidden.to(DEV, BF), src, mod, d["final_layer.linear.weight"], d["final_layer.linear.bias"], out)
    for f, r0 in enumerate(src.tolist()):
        y = R.final_layer(p, hidden[None, r0:r0 + ntok], c[f:f + 1])
        ref = R.unpatchify(y, h, w, 2, 4)
        assert rel_l2(out[f:f + 1], ref) < 1e-2


@pytest.mark.parametrize("pred_type", ["x1", "v"])
@pytest.mark.parametrize("use_cfg", [True, False])
def test_euler_cfg_update(ops, pred_type, use_cfg):
    nf, elems = 4, 4 * 8 * 8
    sigma = R.scheduler_sigma(5, 1.0)
    z0 = torch.randn(nf, elems, generator=g(39))
    if use_cfg:
        z0[nf // 2:] = z0[: nf // 2]
    preds = [bf(torch.randn(nf, elems, generator=g(40 + i))) for i in range(5)]
    z = z0.clone().to(DEV)
    zm = torch.empty(nf, elems, dtype=BF, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    ts = torch.empty(nf, dtype=torch.float32, device=DEV)
    sig_d = sigma.to(DEV)
    for i in range(5):
        ops.sampler_set_timesteps(sig_d, step, ts)
        assert torch.allclose(ts.cpu(), torch.full((nf,), float(sigma[i])))
        ops.euler_cfg_update(z, zm, preds[i].to(DEV, BF), sig_d, step, ops.PRED_X1 if pred_type == "x1" else ops.PRED_V,
                             use_cfg, 1.6)
        ops.sampler_advance(step)
    it = iter(preds)

    def func(zl, t):
        pr = next(it)
        out = [pr[j] for j in range(nf)]
        if use_cfg and pred_type == "v":  # CFG of the 'v' path lives in forward_with_cfg (LVM/model.py:555-562)
            half = nf // 2
            cond = [out[half + j] + 1.6 * (out[j] - out[half + j]) for j in range(half)]
            out = cond + cond
        return out

    ref = R.scheduler_call(sigma, [z0[j] for j in range(nf)], func, use_cfg, 1.6, pred_type)
    assert rel_l2(z, torch.stack(ref)) < 1e-5
    assert rel_l2(zm, torch.stack(ref)) < 4e-3
    assert int(step.item()) == 5


def test_hip_graph_replay(ops):
    """A captured launch sequence replays with updated device-side state (sampler loop under hipGraph)."""
    x = torch.randn(4, 256, device=DEV).to(BF)
    w = torch.ones(256, device=DEV, dtype=BF)
    y = torch.empty_like(x)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        graph = ops.HipGraph().capture(lambda: (ops.rmsnorm(x, w, 1e-5, out=y), ops.sampler_advance(step)))
        for _ in range(3):
            graph.replay()
    s.synchronize()
    assert int(step.item()) == 3
    assert rel_l2(y, R.rmsnorm(x.cpu().float(), w.cpu().float(), 1e-5)) < 4e-3


