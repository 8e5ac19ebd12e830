"""CPU oracle for SURVEY.md row N3 (confidence-service cosines and score statistics) - TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates, with plain Python loops over Python floats (IEEE double, no numpy inside the arithmetic):

* numpy's float64 add-reduction order (numpy/_core/src/umath/loops_utils.h.src, `@TYPE@_pairwise_sum`, numpy 1.x-2.x):
  fewer than 8 elements: left to right from 0.0; 8..128 elements: eight strided partial sums combined as
  ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail (n mod 8 elements) added left to right; more than 128: halves,
  recursively. np.mean = that sum / n; np.var = mean of (x - mean)^2 summed the same way; np.std = sqrt(var).
  Pinned against numpy itself in tests/test_confidence_cpu.py and, through the functions below, against the
  reference-run fixture tests/golden/confidence_cases.json.
* services/multidimensional_confidence_service.py:936-963 `_assess_model_uncertainty`,
  :1087-1099 `_calculate_prediction_variance`, :1101-1114 `_calculate_confidence_interval`,
  :273-280 the `semantic_coherence` cosine (sklearn.metrics.pairwise.cosine_similarity of two float64 rows: each row
  divided by its Euclidean norm, then their dot product; the dot's summation order belongs to the BLAS underneath
  and is NOT pinned: the fixture is matched to 1e-14).
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

PW_BLOCKSIZE = 128


def np_pairwise_sum(a: Sequence[float]) -> float:
    n = len(a)
    if n < 8:
        res = 0.0
        for x in a:
            res += x
        return res
    if n <= PW_BLOCKSIZE:
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return np_pairwise_sum(a[:n2]) + np_pairwise_sum(a[n2:])


def np_sum(a: Sequence[float]) -> float:
    return 0.0 + np_pairwise_sum(a)     # (the reduction starts from the identity)


def np_mean(a: Sequence[float]) -> float:
    return np_sum(a) / len(a)


def np_var(a: Sequence[float]) -> float:
    m = np_sum(a) / len(a)
    d = [(x - m) for x in a]
    return np_sum([x * x for x in d]) / len(a)


def np_std(a: Sequence[float]) -> float:
    return math.sqrt(np_var(a))


def assess_model_uncertainty(scores: List[float]) -> float:
    """:936-963 (scores = [r.get('score', 0) for r in candidate_records])"""
    if not scores:
        return 0.0
    std_score = np_std(scores)
    uncertainty_score = 1.0 - min(std_score, 0.5) / 0.5
    score_confidence = max(scores)
    final_uncertainty = (uncertainty_score * 0.6 + score_confidence * 0.4)
    return min(final_uncertainty, 1.0)


def prediction_variance(scores: List[float]) -> float:
    """:1087-1099"""
    return np_var(scores) if len(scores) > 1 else 0.1


def confidence_interval(confidence: float, variance: float) -> Tuple[float, float]:
    """:1101-1114"""
    margin = 1.96 * math.sqrt(variance)
    return (max(0.0, confidence - margin), min(1.0, confidence + margin))


def semantic_coherence(query_vector: Sequence[float], candidate_vector: Sequence[float]) -> float:
    """:273-280 with sklearn's cosine_similarity spelled out (left-to-right sums: see the module docstring)"""
    nx = math.sqrt(sum(x * x for x in query_vector))
    ny = math.sqrt(sum(y * y for y in candidate_vector))
    nx = nx if nx != 0.0 else 1.0     # sklearn.preprocessing.normalize leaves zero rows alone
    ny = ny if ny != 0.0 else 1.0
    return sum((x / nx) * (y / ny) for x, y in zip(query_vector, candidate_vector))
