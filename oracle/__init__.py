"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU / numpy, fp32) of the reference's detector_translator
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import anything from here; the product package never does.

PARITY UNPINNED: the reference has no tests / golden vectors, and its arithmetic
lives in the un-vendored ``tensorflow-gpu==1.12.0`` (requirements.txt:16) which
cannot be installed here.  The TF-1.12 semantics this restatement encodes
 are assumptions; only the formula / axis conventions of
``utils/model.py`` are pinned against the reference's own file
(tests/golden/make_golden.py).
"""
