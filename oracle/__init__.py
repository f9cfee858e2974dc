"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the one hot path this repository accelerates (embedding-row
gather -> in-batch score matrix + the seven ``xfmr_rec/losses.py`` losses ->
SGD / row-wise Adam update, and exact brute-force top-k retrieval).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import anything from here, and only as the checker / the timed
CPU baseline -- never as part of the product path.  The product
(``matrix-factorization-torch_amd``) never imports this package and raises if its
HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * losses (``oracle.losses``): PINNED -- checked in ``tests/test_oracle_golden.py``
    against ``tests/golden/*.npz``, vectors produced in the build container by
    importing the reference's own ``xfmr_rec/losses.py``
    (``tests/golden/make_golden.py``).
  * embedding tower, SGD / Adam row update, logQ correction, brute-force top-k,
    sharded merge: the reference has no implementation of these,
    so these restatements are OUR spec -- **parity unpinned**.
"""
