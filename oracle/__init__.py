"""CPU oracle for the ADDvisor explanation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker; the product path (``xai-audio-deepfakes_amd/``) never does
and fails loudly if the HIP library is missing.

It is a plain-torch (stock CPU kernels, fp32) restatement of the reference's
arithmetic, each function citing the reference ``file:line`` it follows
(paths relative to the upstream repo davidcombei/xAI-Audio-Deepfakes).

Pinning: the reference has no tests or golden vectors of its own.
``tests/golden/make_golden.py`` therefore runs the reference's *own* modules
(``classifier_embedder``, ``audioprocessor``, ``addvisor``, ``loss_function``) in the
build container -- with their three checkpoint loaders redirected to seeded synthetic
weights, because the real checkpoints are private -- and commits the resulting
input/output vectors under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks
this oracle against every one of them.  Pieces whose reference file cannot be
imported at all (``LMAC_metrics.py``: imports a class that does not exist, SURVEY D1;
``captum_saliency.py``: needs captum; ``hifigan.py``: needs speechbrain) are
**parity unpinned**: the oracle follows the reference text and, for Captum and
HiFi-GAN, the published algorithms.
"""
