#!/usr/bin/env python3
"""Shape statistics of the reference's ICD-10 CSV (data/ICD_10v601.csv) - DATA ONLY, no row content: how many codes of each
hierarchy level there are, how many children a code has, and how long the disease names are (characters) per level; plus
the length distribution of the resulting `semantic_text` (what the encoder sees at build time). scripts/bench_build.py
synthesises a CSV of the same shape from these numbers (the real file cannot travel to the GPU box) and times the full
corpus build on it.

    python tests/golden/make_csv_shape.py      (in the build container: reads /root/reference/data/ICD_10v601.csv)
"""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def hist(values):
    c = collections.Counter(int(v) for v in values)
    return {str(k): c[k] for k in sorted(c)}


def main():
    from rag_project_icd10_amd.tools.build_database import DatabaseBuilder
    recs = DatabaseBuilder().load_csv_data("/root/reference/data/ICD_10v601.csv")   # (pinned to the reference: csv_full_digest.json)
    by_level = collections.defaultdict(list)
    kids = collections.Counter()
    for r in recs:
        by_level[r["level"]].append(len(r["preferred_zh"]))
        if r["parent_code"]:
            kids[(r["level"], r["parent_code"])] += 1
    codes = {r["code"]: r["level"] for r in recs}
    # children per parent, by (parent level, child level); parents without children count as 0
    fan = collections.defaultdict(list)
    for (lv, parent), n in kids.items():
        fan[f"{codes.get(parent, 0)}->{lv}"].append(n)
    childless = {str(l): sum(1 for c, lv in codes.items() if lv == l and not any((k[1] == c) for k in kids)) for l in (1, 2)}
    st = sorted(len(r["semantic_text"]) for r in recs)
    out = {
        "generated_by": "tests/golden/make_csv_shape.py from /root/reference/data/ICD_10v601.csv (statistics only)",
        "rows": len(recs),
        "level_counts": {str(l): len(v) for l, v in sorted(by_level.items())},
        "name_len_hist": {str(l): hist(v) for l, v in sorted(by_level.items())},
        "children_hist": {k: hist(v) for k, v in sorted(fan.items())},
        "parents_missing_from_csv": sum(1 for (lv, p) in kids if p not in codes),
        "childless": childless,
        "combo_codes": sum(1 for r in recs if r["has_complication"]),
        "code_len_hist": hist(len(r["code"]) for r in recs),
        "semantic_text_len": {"mean": sum(st) / len(st), "p50": st[len(st) // 2], "p90": st[int(len(st) * 0.9)], "p99": st[int(len(st) * 0.99)],
                              "max": st[-1], "hist_by_8": hist((x // 8) * 8 for x in st)},
    }
    json.dump(out, open(os.path.join(HERE, "csv_shape.json"), "w"), ensure_ascii=False, indent=0)
    print({k: out[k] for k in ("rows", "level_counts", "parents_missing_from_csv", "childless", "combo_codes")}, out["semantic_text_len"]["mean"],
          {k: len(v) for k, v in out["children_hist"].items()})


if __name__ == "__main__":
    main()
