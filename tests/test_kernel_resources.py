"""No kernel of a shipped library spills registers to scratch (read from the code objects' metadata: runs without a GPU)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_regs  # noqa: E402
from sqair_amd import _capi  # noqa: E402


# (the product and the wide library; the timeline build -- the same sources plus the per-wave stamps -- pushes
# k_insert_loglik_bwd_rows 2-3 registers over its budget and is a measurement tool)
@pytest.mark.parametrize("path", [_capi.LIB_PATH, _capi.WIDE_LIB_PATH])
def test_no_kernel_spills_to_scratch(path):
    if not os.path.exists(path):
        pytest.skip("library not built")
    table = kernel_regs.kernel_table(path)
    assert len(table) > 100, "metadata not parsed"
    bad = [(r["name"], r.get("vgpr_spill_count", 0), r.get("private_segment_fixed_size", 0)) for r in table
           if r.get("vgpr_spill_count", 0) > 0 or r.get("private_segment_fixed_size", 0) > 0]
    assert not bad, bad
