import json

import numpy as np


def rel_max(a, ref):
    """max |a - ref| / max |ref|  (norm-relative, the criterion BASELINE.md section 4 states)."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - ref)) / max(float(np.max(np.abs(ref))), 1e-30))


def ulp_diff(a, b):
    """Distance in float32 units-in-the-last-place, elementwise (signed-magnitude ordering)."""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)


def meta_of(npz):
    return json.loads(str(npz["meta"]))


# tolerances of BASELINE.json north_star: state <= 1e-5 relative, gradients <= 1e-4 (norm-relative)
TOL_STATE = 1e-5
TOL_GRAD = 1e-4
