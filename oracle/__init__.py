"""CPU oracle for the JyutVoice hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32 restatement of the reference's algorithm for the path in SURVEY.md 8(a)
(text encoder -> duration/length regulation -> CFM Euler/CFG loop over the flow estimator -> HiFT
vocoder), written functionally over a flat state-dict so it shares no structure with the product.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and
only as the checker / the reported CPU baseline -- never as the thing measured or shipped.  The
product (jyutvoice_amd/) does not import it and fails loudly when the HIP library is missing.

Pinning: the reference has no tests or golden vectors for this path.  The oracle
is pinned by tests/golden/*.npz, generated in the build container by tests/golden/make_golden.py,
which imports the reference's own source files from /root/reference, loads the key-hashed
synthetic weights into the reference's modules, and records their outputs:
  * rows a3, a4, a5, a9 (text encoder, duration predictor, length regulation, HiFT): genuine
    imported-reference outputs -> parity PINNED.
  * rows a6-a8 (CFM loop, estimator, transformer block): the reference's decoder.py /
    transformer.py / flow_matching.py run unmodified, but their attention/GELU arithmetic lives in
    the third-party `diffusers==0.35.2` (requirements.txt:1), which is absent here.  make_golden.py
    supplies restated `Attention`/`GELU` classes following diffusers 0.35.2's published
    AttnProcessor2_0 semantics (q/k/v Linear no-bias, heads=8 x 64, additive mask broadcast over
    heads, F.scaled_dot_product_attention scale 1/sqrt(64), to_out Linear+bias; GELU = exact-erf
    gelu(Linear(x))).  Parity at that boundary is therefore UNPINNED numerically and pinned only
    structurally (910 tensors / 71 302 480 parameters, README.md:171,233).
"""
