"""oracle/ -- CPU restatement of the reference's algorithm for the denoiser hot path.

TEST INFRASTRUCTURE ONLY. Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
Nothing under dimsum_amd/ imports this package; the product path fails loudly without the HIP library.

Parity status: PINNED -- tests/test_oracle_golden.py checks every function against golden vectors captured from
the reference's own pure-PyTorch paths in the build container (tools/gen_golden.py; fixtures in tests/golden/).
Third-party pieces with no reference-side pin are listed in DESIGN.md ("parity unpinned": torchdiffeq stepping).

  ssm_oracle.c   C (double-precision internals): selective scan fwd/bwd, causal conv1d fwd/bwd, RMS/LayerNorm
  c_ops.py       ctypes binding of the above on numpy arrays
  np_ops.py      numpy: scan-order tables, block reorders, Haar/DCT 4x4 token transforms, fusion attention,
                 gated GeLU, sinusoidal embeddings
"""
