"""CPU oracle for the TIC fine-tune hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
The product package (``touhouimageclassification_amd``) never imports it and
fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * ViT   : pinned against HuggingFace ``ViTForImageClassification`` built from a
            local ``ViTConfig`` (transformers 5.15.0 -- the third-party library the
            reference delegates its ViT arithmetic to, TIC/ViT/model.py:2,27-45)
            by tools/gen_golden.py -> tests/golden/vit_*.npz.
  * ResNet: pinned against the reference's own TIC/ResNet/model.py, imported in
            the authoring container by tools/gen_golden.py -> tests/golden/resnet_*.npz.
  * Aug / MixUp / CutMix: torchvision is absent and the reference holds no test
            for them -> "parity unpinned" (restated from documented v2 behaviour).
"""
