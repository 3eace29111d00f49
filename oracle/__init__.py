"""TEST INFRASTRUCTURE ONLY.

CPU restatement ("oracle") of the reference hot path (PatrickHwang/Explicit-tf2-Recommendation,
``2.FM/CustomLayers.py``, ``3.DCN/CustomLayers.py``, ``5.DIN/CustomLayers.py`` and the train step of
``2.FM/ModelManager.py``).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package -- and only as the checker / the timed CPU baseline,
never as part of the shipped HIP path.

Pinning status:
  * DSSM towers (gather addressing, field-major flatten, x@K+b, relu-on-every-hidden-layer, linear
    final layer) are PINNED against the reference's own artifacts: ``ckpt-7`` weights reproduce the
    vectors in ``ebd_result/{user,item}_embedding.json`` (tests/golden/dssm_ckpt7_kat.npz).
  * FM, DeepFM, CrossNet (vector and matrix), DIN ActivationUnit, cosine, BCE, Adam have no
    reference-held expected outputs (the reference has no tests and TensorFlow cannot be imported
    here: ModuleNotFoundError, nothing refused): for these **parity is unpinned**.  They are held by
    two independent restatements that must agree (``layers_np`` closed-form numpy with derived
    backward vs ``torch_ref`` op-for-op torch-CPU with autograd).
"""
