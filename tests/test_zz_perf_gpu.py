"""Wall-clock limits on what the GPU parity tests measured (tests/perf_records.py). Collected LAST (file name), so that a slow
or noisy box cannot stop `pytest -x` in front of a parity test. Two tiers:
  * HARD limits, 5-10 x what a quiet box shows: a real failure - a performance regression of that size is a bug, not box noise;
  * TIGHT limits, 2-4 x the quiet-box figures: `xfail(strict=False)` - a slow box shows up as XFAIL in the summary
    (scripts/gpu_final.sh prints the -rx lines), never as a failure that masks parity results."""
import pytest

import perf_records

pytestmark = pytest.mark.gpu

# name suffix -> (statistic, tight limit in ms, hard limit in ms); tests/test_gpu_parity.py prints the figures
LIMITS = {
    "_fresh_index_batches_ms": [("min", 2.0, 10.0), ("median", 4.0, 20.0)],   # device time of a batch = the fastest of nine (0.8-1.4 ms quiet)
    "_first_large_batch_ms": [("min", 12.0, 40.0)],                            # ONE measurement: 2.2-2.4 ms on a quiet host
    "_wide_mode_batches_ms": [("median", 3.0, 12.0)],                          # 1.4-1.5 ms quiet
}


def _stat(kind, v):
    v = sorted(v)
    return v[0] if kind == "min" else v[len(v) // 2]


def _over(tier):
    bad = []
    for name, values in perf_records.RECORDS.items():
        for suffix, checks in LIMITS.items():
            if name.endswith(suffix):
                for kind, tight, hard in checks:
                    limit = tight if tier == "tight" else hard
                    got = _stat(kind, values)
                    if got > limit:
                        bad.append(f"{name}: {kind} {got:.2f} ms > {limit} ms ({values})")
    return bad


def test_recorded_wall_clock_figures_are_within_hard_limits():
    if not perf_records.RECORDS:
        pytest.skip("no parity test recorded a timing in this process")
    bad = _over("hard")
    assert not bad, bad


@pytest.mark.xfail(strict=False, reason="tight wall-clock limit on a shared box")
def test_recorded_wall_clock_figures_are_within_tight_limits():
    if not perf_records.RECORDS:
        pytest.skip("no parity test recorded a timing in this process")
    bad = _over("tight")
    assert not bad, bad
