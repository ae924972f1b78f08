"""CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain NumPy (float32, explicit formulas, no autograd) restatement of the
reference's algorithm for the hot path named by BASELINE.json:north_star
(twndus/YelpRecommendation: BPR-MF, NGCF propagation, CDAE, ranking metrics).
Every function cites the reference file:line it follows.

Who may import this package: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` — as the checker or the reported CPU
baseline, never as the thing shipped.  Nothing under ``yelprecommendation_amd/``
imports it; the product path raises if the HIP library is missing.

Pinning: the arithmetic of this path lives in a third-party dependency of the
reference (torch ATen, pinned ``torch==2.2.2`` in poetry.lock:2187-2188; not under
/root/reference) and the reference's own tests hold vectors only for
``metric.py`` (test/test_metric.py:9-47).  The oracle is therefore pinned by
(a) those four known-answer tests and (b) golden vectors captured by running the
reference itself in the build container (``tests/golden/make_golden.py`` is the
generating script; torch 2.10 / numpy 2.2 / sklearn 1.7 — versions are stored in
each fixture).  ``tests/test_oracle_*.py`` check every oracle function against
those fixtures.  ``cdae_batches.py`` (the CPU statement of the two CDAE batch kernels) is pinned
differently: its dense rows against the reference-style pipeline's masks, its negative masks
through the law of ``np.random.choice(.., replace=False)`` (tests/test_cdae_batches.py) — the random
stream itself cannot be the reference's.
"""
