"""CPU oracle for the AdvShadow hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
directory: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

Every function restates, in plain torch-CPU / numpy arithmetic, one piece of
the reference's algorithm and cites the reference ``file:line`` it follows.
The restatement is pinned against the reference itself: ``tests/golden/``
holds vectors produced by importing ``/root/reference`` in the build container
(``tests/golden/make_golden.py``), and ``tests/test_oracle_*.py`` compares the
oracle with them.  Pieces whose third-party arithmetic is absent from the image
(cv2, skimage, timm/torchvision victims) say "parity unpinned" in their
docstrings.
"""
