"""Wall-clock limits on what the GPU parity tests measured (tests/perf_records.py). Collected LAST (file name), and
`xfail(strict=False)`: a slow box shows up as XFAIL in the summary, never as a failure that masks parity results."""
import pytest

import perf_records

pytestmark = pytest.mark.gpu

# name prefix -> (statistic, limit in ms); the limits are 2-4x what a quiet box shows (tests/test_gpu_parity.py prints the figures)
LIMITS = {
    "_fresh_index_batches_ms": [("min", 2.0), ("median", 4.0)],     # device time of a batch = the fastest of nine
    "_first_large_batch_ms": [("min", 12.0)],                        # ONE measurement: 2.2-2.4 ms on a quiet host
    "_wide_mode_batches_ms": [("median", 3.0)],
}


def _stat(kind, v):
    v = sorted(v)
    return v[0] if kind == "min" else v[len(v) // 2]


@pytest.mark.xfail(strict=False, reason="wall-clock limit on a shared box")
def test_recorded_wall_clock_figures_are_within_limits():
    if not perf_records.RECORDS:
        pytest.skip("no parity test recorded a timing in this process")
    bad = []
    for name, values in perf_records.RECORDS.items():
        for suffix, checks in LIMITS.items():
            if name.endswith(suffix):
                for kind, limit in checks:
                    got = _stat(kind, values)
                    if got > limit:
                        bad.append(f"{name}: {kind} {got:.2f} ms > {limit} ms ({values})")
    assert not bad, bad
