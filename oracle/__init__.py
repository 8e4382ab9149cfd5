"""CPU oracle for the garage on-policy hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / scipy / torch-CPU) of the
algorithm the reference (akolobov/garage v2021.03.0) runs on its
``LocalSampler``/``VecWorker`` rollout -> returns + GAE(lambda) ->
``PPO._train_once`` path.  Every function cites the reference file:line it
follows (paths relative to ``/root/reference/src/garage`` unless they start
with ``tests/``).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / reported baseline,
never as the thing measured or shipped.  ``garage_amd`` (the product) must
never import anything from here; ``tests/test_no_oracle_in_product.py``
enforces that.

Parity status: PINNED.
  * against the reference's own literal test vectors
    (tests/garage/torch/test_functions.py:86-117,
    tests/garage/test_functions.py:49-97, tests/garage/np/test_functions.py:81-88,
    tests/garage/test_dtypes.py:238-318,
    tests/garage/torch/modules/test_gaussian_mlp_module.py:98-123), and
  * against outputs of the real reference code executed in the build
    container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
"""
