python - <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from test_gpu_parity import _family_corpus
from conftest import icd_levels
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
corpus, queries = _family_corpus(300, 120, 768, 77)
levels = icd_levels(len(corpus), 78)
for env in (None, "1"):
    if env: os.environ["ICD_NO_PERMUTE"] = env
    idx = IcdIndex(corpus, levels, max_nq=256, max_k=10)
    idx.search(queries, 10, MODE_AUTO)
    print("ICD_NO_PERMUTE=%s  fallback %d of 256  lists %d" % (env, idx.stats()["last_fallback"], idx.stats()["last_chunks"]))
    idx.close()
PY
bash scripts/gpu_check.sh
