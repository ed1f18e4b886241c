"""CPU oracle for the Lifelong-nnUNet hot path -- TEST INFRASTRUCTURE ONLY.

This package is a pure-PyTorch, CPU, fp32 restatement of the reference's training step and
continual-learning regularisers.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product package
(``lifelong-nnunet_amd/``) never imports anything from here and fails loudly when its HIP
library is missing.

Parity pinning (see DESIGN.md "Oracle"):
  * ``oracle.losses`` EWC / LwF terms are validated against the reference's own
    ``nnunet_ext/training/loss_functions/deep_supervision.py`` classes executed verbatim in the
    build container through a 3-symbol shim (``oracle/make_goldens.py``); the resulting values are
    committed under ``tests/golden/``.
  * ``oracle.unet`` / Dice+CE / the iteration restate *upstream* nnU-Net v1
    (nnunet @ 77bc485ee025a61feb91cb0a0ed1c61a32a0f39f, requirements.txt:4), whose source is not
    under /root/reference and is not installed.  The reference's own tests hold no numeric golden
    for it, so that part is "parity unpinned by the reference"; it is pinned against the
    structural evidence the reference does hold (module tree dump, ctor args, forward copy).
"""
