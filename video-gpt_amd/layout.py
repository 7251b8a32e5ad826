"""TokenLayout: the attention mask of a batch of clip sequences as 8 bytes per TOKEN instead of one byte per
(query, key) pair.

The reference's collator paints a dense (B, L, L) bool mask on the host (LVM/processor.py:575-731: 9.6 MB per row
at L=3096, 1 GB per row at L=31 806), the model turns it into an additive (B,1,L,L) tensor on every forward
(OmniGen/transformer.py:139-145).  Every mask those builders produce follows one rule over per-token attributes
(processor.py module docstring), so the product path keeps only the attributes, transforms THEM when the engine
re-lays the sequence out (left pads dropped and rows concatenated; a gap inserted behind the condition prefix), and
lets the device expand them straight into the bit-packed rows and the tile summary the attention kernel reads
(include/vgpt.h, vgpt_mask_build_tokens) — SURVEY.md §8f.3.  `to_bool()` is the dense form for callers and tests
that want the reference's tensor; tests/test_layout.py pins it bit-for-bit against the golden collator masks.

Rule (q = query row, k = key column, both indices into the row's L tokens):
    kind[q] == PAD                                   -> visible            (pad rows see everything, :726-727)
    kind[k] == CLEAN and seq[q] == seq[k]            -> visible iff q >= thr[k]
    kind[k] == NOISY and kind[q] == NOISY
                     and grp[q] == grp[k]            -> visible iff oc[q] >= oc[k] and (sub[k] == 0 or sub[k] == sub[q])
    otherwise (PAD / GAP key, other sequence, ...)   -> masked
with thr[k] = k for `<img>` / `</img>`, block_start + 1 for image slots; oc = min(in-block offset, 2); grp unique per
(sequence, clip); GAP tokens (alignment filler the engine inserts) see nothing and are seen by nothing.  `sub` is 0 on
every token the collator lays out; the engine's per-clip pass (engine.StaticDenoiser._clip_pass) numbers the time rows of
denoise step s with sub = s + 1 so that ONE sequence holds the time tokens of every step: they all see their clip's
`<|diffusion|>` columns (sub 0) and only the time columns of their own step.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

PAD, CLEAN, NOISY, GAP = 0, 1, 2, 3
GAP_SEQ = 255           # sequence id no real token carries
MAX_SEQ = 255
MAX_CLIPS = 4096        # clips per sequence (grp = seq * MAX_CLIPS + clip)


class TokenLayout:
    def __init__(self, thr, seq, kind, oc, grp, sub=None):
        self.thr, self.seq, self.kind, self.oc, self.grp = (np.ascontiguousarray(a, dtype=np.int64)
                                                            for a in (thr, seq, kind, oc, grp))
        self.sub = np.zeros_like(self.thr) if sub is None else np.ascontiguousarray(sub, dtype=np.int64)
        if not (self.thr.ndim == 2 and self.thr.shape == self.seq.shape == self.kind.shape == self.oc.shape == self.grp.shape
                == self.sub.shape):
            raise ValueError("TokenLayout: attribute arrays must all be (B, L)")
        if self.sub.min(initial=0) < 0 or self.sub.max(initial=0) >= 1 << 24 or bool((self.sub[self.kind != NOISY] != 0).any()):
            raise ValueError("TokenLayout: sub must be in [0, 2^24) and 0 on every token that is not NOISY")
        if self.thr.shape[1] >= 1 << 24:
            raise ValueError("TokenLayout: L must be below 2^24")
        self._pm = {}

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def from_plans(cls, plans: Sequence[Tuple[Sequence[Tuple[int, int]], int, int]], L: int) -> "TokenLayout":
        """plans[b] = (kinds, bl, pad): the block plan of row b (processor.plan_*: (kind, clip id) per block), its
        block length and its left-pad length; len(kinds) * bl + pad must equal L."""
        B = len(plans)
        if B > MAX_SEQ:
            raise ValueError("TokenLayout: at most 255 rows")
        thr, seq, kind, oc, grp = (np.zeros((B, L), dtype=np.int64) for _ in range(5))
        for b, (kinds, bl, pad) in enumerate(plans):
            n = len(kinds)
            if n * bl + pad != L:
                raise AssertionError("block plan does not cover the valid tokens")
            seq[b] = b
            if n == 0:
                continue
            kd = np.repeat(np.array([k for k, _ in kinds], dtype=np.int64), bl)
            cl = np.repeat(np.array([c for _, c in kinds], dtype=np.int64), bl)
            if cl.max(initial=0) >= MAX_CLIPS:
                raise ValueError("TokenLayout: too many clips in one row")
            off = np.tile(np.arange(bl, dtype=np.int64), n)
            start = pad + np.repeat(np.arange(n, dtype=np.int64) * bl, bl)
            pos = pad + np.arange(n * bl, dtype=np.int64)
            kind[b, pad:] = kd
            oc[b, pad:] = np.minimum(off, 2)
            edge = (off == 0) | (off == bl - 1)
            thr[b, pad:] = np.where(kd == CLEAN, np.where(edge, pos, start + 1), 0)
            grp[b, pad:] = np.where(kd == NOISY, b * MAX_CLIPS + cl, 0)
        return cls(thr, seq, kind, oc, grp)

    # ---- shape / device plumbing ----------------------------------------------------------------------------------
    @property
    def B(self) -> int:
        return self.thr.shape[0]

    @property
    def L(self) -> int:
        return self.thr.shape[1]

    @property
    def shape(self):
        return (self.B, self.L, self.L)

    def dim(self) -> int:
        return 3

    def size(self, i: int) -> int:
        return self.shape[i]

    def to(self, *_, **__):
        """Host-side description; the device form is built by packed_mask()."""
        return self

    # ---- the dense mask (reference tensor form) ---------------------------------------------------------------------
    def to_bool(self) -> np.ndarray:
        B, L = self.B, self.L
        out = np.zeros((B, L, L), dtype=np.bool_)
        q = np.arange(L, dtype=np.int64)[:, None]
        for b in range(B):
            kq, kk = self.kind[b][:, None], self.kind[b][None, :]
            vis = (kk == CLEAN) & (self.seq[b][:, None] == self.seq[b][None, :]) & (q >= self.thr[b][None, :])
            vis |= (kk == NOISY) & (kq == NOISY) & (self.grp[b][:, None] == self.grp[b][None, :]) & \
                   (self.oc[b][:, None] >= self.oc[b][None, :]) & \
                   ((self.sub[b][None, :] == 0) | (self.sub[b][None, :] == self.sub[b][:, None]))
            vis |= (kq == PAD)
            out[b] = vis
        return out

    def to_bool_tensor(self):
        import torch
        return torch.from_numpy(self.to_bool())

    # ---- re-layouts the engine applies ----------------------------------------------------------------------------
    def left_pads(self) -> List[int]:
        """Left-pad length of every row (leading PAD tokens)."""
        lead = np.cumprod(self.kind == PAD, axis=1).sum(1)
        return [int(v) for v in lead]

    def pack(self, pads: Optional[List[int]] = None):
        """Drop every row's left pads and lay all rows out as ONE sequence (engine.pack_left_padded): tokens keep
        their sequence id, so the mask of the packed row is block-diagonal by construction.
        Returns (layout with B == 1, offsets of the rows in the packed sequence)."""
        pads = self.left_pads() if pads is None else pads
        offsets, o = [], 0
        cols = {n: [] for n in ("thr", "seq", "kind", "oc", "grp", "sub")}
        for b in range(self.B):
            p = pads[b]
            offsets.append(o)
            cols["thr"].append(self.thr[b, p:] - p + o)
            for n in ("seq", "kind", "oc", "grp", "sub"):
                cols[n].append(getattr(self, n)[b, p:])
            o += self.L - p
        return TokenLayout(*(np.concatenate(cols[n])[None, :] for n in ("thr", "seq", "kind", "oc", "grp", "sub"))), offsets

    def insert_gap(self, at: int, n: int) -> "TokenLayout":
        """n GAP tokens in front of position `at` of every row (the engine aligns the static condition prefix to the
        attention kernel's 128-row blocks); visibility thresholds behind the gap move with their tokens."""
        if n == 0:
            return self
        B = self.B

        def ins(a, fill):
            return np.concatenate([a[:, :at], np.full((B, n), fill, dtype=np.int64), a[:, at:]], axis=1)
        thr = np.where(self.thr >= at, self.thr + n, self.thr)
        return TokenLayout(ins(thr, 0), ins(self.seq, GAP_SEQ), ins(self.kind, GAP), ins(self.oc, 0), ins(self.grp, 0),
                           ins(self.sub, 0))

    def permute(self, perm) -> "TokenLayout":
        """Tokens re-ordered: new position i holds old token perm[i]; perm[i] == -1 inserts a GAP token.  Only valid
        when every CLEAN token keeps its position (their visibility is a threshold in sequence order) and every
        non-CLEAN token stays behind all CLEAN ones — the engine moves NOISY rows among themselves."""
        perm = np.asarray(perm, dtype=np.int64)
        keep = perm >= 0
        src = np.where(keep, perm, 0)
        clean_old = np.nonzero((self.kind == CLEAN).any(axis=0))[0]
        if clean_old.size:
            last = int(clean_old.max())
            if not np.array_equal(perm[: last + 1], np.arange(last + 1)):
                raise ValueError("TokenLayout.permute: CLEAN tokens (and everything before the last one) must stay in place")

        def take(a, fill):
            return np.where(keep[None, :], a[:, src], fill)
        return TokenLayout(take(self.thr, 0), take(self.seq, GAP_SEQ), take(self.kind, GAP), take(self.oc, 0),
                           take(self.grp, 0), take(self.sub, 0))

    def with_groups(self, grp) -> "TokenLayout":
        """Same tokens with other clip-group ids for the NOISY ones."""
        return TokenLayout(self.thr, self.seq, self.kind, self.oc, grp, self.sub)

    def with_subgroups(self, sub) -> "TokenLayout":
        """Same tokens with other sub-group numbers for the NOISY ones (engine: time rows of step s carry s + 1)."""
        return TokenLayout(self.thr, self.seq, self.kind, self.oc, self.grp, sub)

    def prefix_is_static(self, t_first: int, row: int = 0) -> bool:
        """True when no row before t_first can see a column at or behind it — the rows of the condition prefix are
        then the same at every denoise step."""
        kq = self.kind[row, :t_first]
        if (kq == PAD).any():
            return False
        kk, thr, seq = self.kind[row, t_first:], self.thr[row, t_first:], self.seq[row, t_first:]
        clean_seen = (kk == CLEAN) & (thr < t_first) & np.isin(seq, self.seq[row, :t_first][kq != GAP])
        if clean_seen.any():
            return False
        noisy_q = kq == NOISY
        if noisy_q.any() and np.isin(self.grp[row, t_first:][kk == NOISY], self.grp[row, :t_first][noisy_q]).any():
            return False
        return True

    # ---- device form ------------------------------------------------------------------------------------------------
    def attr(self) -> np.ndarray:
        """(B, L, 2) int32: word 0 = (thr of a CLEAN token, sub of a NOISY one, else 0) | seq << 24,
        word 1 = kind | oc << 2 | grp << 4 (include/vgpt.h)."""
        low = np.where(self.kind == CLEAN, self.thr, np.where(self.kind == NOISY, self.sub, 0)) & 0xFFFFFF
        w0 = (low | (self.seq << 24)).astype(np.uint32)
        w1 = (self.kind | (self.oc << 2) | (self.grp << 4)).astype(np.uint32)
        return np.stack([w0, w1], axis=-1).view(np.int32)

    def packed_mask(self, device):
        """Bit-packed rows + tile summary, expanded on the device from the attributes (no (B,L,L) tensor anywhere)."""
        import torch
        from . import ops
        key = str(device)
        pm = self._pm.get(key)
        if pm is None:
            pm = self._pm[key] = ops.build_mask_from_layout(torch.from_numpy(self.attr()).to(device), self.B, self.L)
        return pm
