"""CPU oracle for the AudioLab Process->Separate hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker.  The product path
(``audiolab_amd``) never imports this package and fails loudly when the HIP
extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * mdx_oracle   (STFT / iSTFT / margin chunker)  -- PINNED against the imported
    in-tree reference ``modules/rvc/infer/modules/uvr5/mdxnet.py`` through the
    golden fixtures in ``tests/golden`` (generator: ``oracle/make_golden.py``).
  * ensemble_oracle (blend / residual subtract / de-bleed) -- PINNED against the
    imported ``modules/separator/stem_separator.py`` helpers (same fixtures).
  * OLA chunker, normalise(0.9), compensate, spectral inversion, ConvTDFNet
    topology -- PARITY UNPINNED: they live in the un-vendored third-party package
    ``audio-separator[gpu]>=0.32.0`` (setup.sh:96), absent from /root/reference
    and not installed; restated from its published algorithm.
"""
