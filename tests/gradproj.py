"""<gradient, fixed pseudo-random direction> per variable: the gradient check of the reference-graph fixture that a norm cannot give.

``tests/golden/make_networks_golden.py`` stores, for every variable of the reference's D / G ``var_list``, the dot product of its
gradient with ``projection_vector(name, shape)``; the oracle (tests/test_reference_graph.py) and the HIP path
(tests/test_model_gpu.py) are held to those numbers.  A sign flip, a transposed or permuted filter gradient leaves the norm
unchanged and moves the projection by ~|g|.
"""
import zlib

import numpy as np


def projection_vector(name, shape):
    """Direction for variable ``name``: standard-normal draws from RandomState(CRC-32 of the name), float64, C order."""
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff).standard_normal(tuple(int(s) for s in shape))


def projection(name, grad):
    g = np.asarray(grad, np.float64)
    return float((g * projection_vector(name, g.shape)).sum())
