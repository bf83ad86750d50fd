"""CPU oracle for the RobustVLM adversarial hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (torch-fp32 / numpy / C) restatement of the
reference algorithm for the path BASELINE.json's ``north_star`` names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker / the timed CPU baseline - never as part of the product path (``robustvlm_amd``), which
fails loudly when the HIP library is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against golden
vectors under tests/golden/ that were produced by running the reference's own code
(/root/reference, imported or AST-extracted - see tests/golden/make_golden.py) in the build
container.  The ViT arithmetic lives in a third-party dependency that is absent from
/root/reference (open-clip-torch==2.19.0, requirements.txt:91); it is restated from its published
architecture and pinned against the independent HF ``transformers``
CLIPVisionModelWithProjection implementation of the same model (tests/golden/vit_hf_*.npz).
"""
