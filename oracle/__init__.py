"""CPU oracle for the vit-tf feature-volume + similarity hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the reported CPU
baseline -- never as the thing measured or shipped.  The product package
(``vit-tf_amd/``) must not import from here and fails loudly when its HIP
library is missing.

Everything is plain PyTorch fp32 on the CPU (the reference's own CPU path is
``dev=cpu, typ=float32``, infer.py:308-309, predict_ntf.py:114-117).

Pinning status (see DESIGN.md "Oracle"):

* ``feature_volume`` / ``similarity`` / ``synthetic`` restate code that lives
  in /root/reference and are pinned by golden vectors produced by running the
  reference's own functions in the build container
  (``tests/golden/make_golden.py``).
* ``dino_vit`` restates a third-party dependency that is NOT vendored in the
  reference (``torch.hub.load('facebookresearch/dino:main', 'dino_vits8')``,
  infer.py:42-43 -- an unpinned moving branch, no network here).  Its parity
  is anchored on the reference's call sites: the hook position
  ``blocks[-1].attn.qkv`` (infer.py:135), ``attn.num_heads`` (infer.py:180)
  and the K-column slicing (infer.py:189-203), all exercised through the
  imported reference harness when the goldens are generated.  The ViT
  arithmetic itself is "parity unpinned" against upstream weights.
"""
