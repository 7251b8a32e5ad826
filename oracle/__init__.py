"""CPU oracle for the Video-GPT next-clip diffusion hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (torch fp32 / numpy) restatement of the
reference algorithm, written from the reference's behaviour with each function citing the
reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it, and only as the checker — never as the thing measured or
shipped.  The product path (`video-gpt_amd/`) never imports it and fails loudly without the HIP
library.

Pinning status (see DESIGN.md "Oracle"):
  * collator / position / mask builders, LVMScheduler, TimestepEmbedder, FinalLayer, PatchEmbedMR,
    sincos tables: PINNED against the reference's own classes, executed in the build container by
    AST extraction (`oracle/extract_reference.py`), vectors committed under tests/golden/.
  * Phi3 decoder layer (RMSNorm, RoPE, attention, gated MLP): the reference calls the un-vendored
    transformers==4.47.1; cross-checked against the installed transformers 5.15 Phi3 classes
    (same formulas).  The reference holds no fixture for it -> parity unpinned by the reference.
  * LVM.frame_block_forward glue, loss, pipeline: restated line by line; the reference holds no
    tests or golden vectors for them -> parity unpinned.
  * VAE (diffusers==0.29.0 AutoencoderKL, absent from the container): restated from the published
    architecture -> parity unpinned.
"""
