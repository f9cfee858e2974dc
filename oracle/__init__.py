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
  * embedding tower, SGD / Adam row update, brute-force top-k, retrieval metrics, the two key
    orders, the MovieLens split / history logic: the reference has no implementation of these
    (SURVEY.md 0.3), so they are pinned against THIRD-PARTY definitions instead
    (``tests/test_oracle_pins.py``, ``tests/test_host_cpu.py``: torch.optim.SGD / AdamW on the
    touched rows, F.embedding + F.normalize, fp64 matmul + stable sort, hand-worked
    torchmetrics-formula cases, an independent restatement of the key orders and of the polars
    expressions).
  * still OUR spec (**parity unpinned**): the logQ correction, the hash towers' combination
    rule, the sharded merge, the batch producer's shuffle, ``init_rows``.
"""
