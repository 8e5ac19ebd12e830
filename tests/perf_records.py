"""Wall-clock figures that parity tests measure on the way are RECORDED here and judged in tests/test_zz_perf_gpu.py,
which pytest collects last: under `pytest -m gpu -x` a slow or noisy box then cannot stop the run in front of a parity
test. Plain module state: one pytest process."""
RECORDS = {}


def record(name, values_ms):
    RECORDS[name] = [float(v) for v in values_ms]
