"""The backend's history scan (csrc/host/block_metric.h): every instruction-set variant this CPU has against the plain C++ loop, on inputs the
whole-backend tests do not reach -- distances up to the metric's maximum, limits from zero to beyond every error, the selector-SAD test on and
off, histories that are unrelated to / close to / equal to the block's own pattern (ties: the first index wins)."""
import pytest

from helpers import block_metric_host


@pytest.mark.parametrize("magnitude", [8, 16, 26])
def test_history_scan_variants_agree(magnitude):
    L = block_metric_host()
    have = L.bm_variants()
    if have == 1:
        pytest.skip("no SIMD variant on this CPU")
    assert L.bm_scan_check(1234 + magnitude, 20000, magnitude, have) == 0


def test_fused_block_search_variants_agree():
    """search_prepare + search_history (pixels -> distance table -> own error -> limit; own-pattern look-up -> filter -> exact sums): the two calls the selector walk makes per block."""
    L = block_metric_host()
    have = L.bm_variants()
    if have == 1:
        pytest.skip("no SIMD variant on this CPU")
    assert L.bm_search_check(99, 40000, have) == 0
