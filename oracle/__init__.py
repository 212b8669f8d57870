"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (torch-CPU / numpy, fp32) of the reference's detector_translator
hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import anything from here; the product package never does.

Pinning status:

* WIRING PINNED to the reference's own files: tests/golden/networks_ref.npz holds what
  models/networks/{__init__,layers,vgg}.py, utils/model.py and
  models/detector_translator_model.py produced when executed unmodified under the lazy
  TensorFlow stand-in tests/golden/tf_standin.py (variable registry and creation order,
  a forward pass, two train steps, a test step); tests/test_reference_graph.py checks
  this restatement against it.  utils/model.py alone is also pinned by
  tests/golden/model_utils_ref.npz, the input pipelines by image_pair_ref.npz.
* OP NUMERICS UNPINNED: the reference has no tests / golden vectors and its arithmetic
  lives in the un-vendored ``tensorflow-gpu==1.12.0`` (requirements.txt:16), which
  cannot be installed here.  The TF-1.12 op semantics encoded here and in the stand-in
  (SAME-pad split, legacy bilinear, fused batch norm, linspace,
  softmax, ApplyAdam) are assumptions.
"""
