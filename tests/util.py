import json

import numpy as np


def rel_max(a, ref):
    """max |a - ref| / max |ref|  (norm-relative, the criterion BASELINE.md section 4 states)."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(a - ref)) / max(float(np.max(np.abs(ref))), 1e-30))


def rel_elem(a, ref, floor=1e-6):
    """Element-wise relative error max_i |a_i - ref_i| / max(|ref_i|, floor * max|ref|): the north_star's "state <= 1e-5
    relative" read entry by entry, with an absolute floor so that entries that are (nearly) zero do not divide by zero."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    return float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref), floor * scale)))


def grad_report(tag, a, ref):
    """Print the per-entry gradient error percentiles (norm-relative and element-wise) so that the gap between the two
    criteria is visible in every test log; returns the norm-relative maximum."""
    a, ref = np.asarray(a, dtype=np.float64).ravel(), np.asarray(ref, dtype=np.float64).ravel()
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    e_norm = np.abs(a - ref) / scale
    e_elem = np.abs(a - ref) / np.maximum(np.abs(ref), 1e-6 * scale)
    q = (50, 90, 99, 100)
    print("%s: |dg| / max|g| p50/p90/p99/max = %s ; element-wise = %s" % (
        tag, " ".join("%.1e" % np.percentile(e_norm, x) for x in q), " ".join("%.1e" % np.percentile(e_elem, x) for x in q)))
    return float(e_norm.max())


def state_report(tag, a, ref):
    """rel_max with the measured value printed (pytest -s): the per-step queue terms / states of the network tests, so that the
    test logs show how far inside the tolerance a run is."""
    e = rel_max(a, ref)
    print("%s: max |d| / max |ref| = %.2e" % (tag, e))
    return e


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
