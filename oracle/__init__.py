"""CPU restatement of the reference's hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package never does (tests/test_boundary.py::test_product_does_not_touch_oracle checks).
"""
from .oracle import *  # noqa: F401,F403
