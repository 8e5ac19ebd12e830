"""Row a20 (build_full_database, CLI flags, exit code) and the `.env` loading of the stand-alone entry points, without a
GPU: the orchestration runs over recording stand-ins for the two services (the GPU test drives the real ones)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from rag_project_icd10_amd import dotenv_lite
from rag_project_icd10_amd.tools import build_database as bd


def test_dotenv_subset(tmp_path, monkeypatch):
    (tmp_path / ".env").write_text(
        "# 向量化配置\nEMBEDDING_MODEL_NAME=shibing624/text2vec-base-chinese\nexport MILVUS_DB_PATH=\"./db/x y.db\"\n"
        "MILVUS_COLLECTION_NAME='icd10_collection'   \nEMPTY=\nINLINE=abc # comment\nALREADY=from_file\nnot a line\n", encoding="utf-8")
    vals = dotenv_lite.parse_dotenv((tmp_path / ".env").read_text(encoding="utf-8"))
    assert vals == {"EMBEDDING_MODEL_NAME": "shibing624/text2vec-base-chinese", "MILVUS_DB_PATH": "./db/x y.db",
                    "MILVUS_COLLECTION_NAME": "icd10_collection", "EMPTY": "", "INLINE": "abc", "ALREADY": "from_file"}
    sub = tmp_path / "a" / "b"
    sub.mkdir(parents=True)
    assert dotenv_lite.find_dotenv(str(sub)) == str(tmp_path / ".env")          # nearest parent directory
    for k in vals:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("ALREADY", "from_env")
    monkeypatch.chdir(sub)
    assert dotenv_lite.load_dotenv() is True
    assert os.environ["MILVUS_COLLECTION_NAME"] == "icd10_collection" and os.environ["ALREADY"] == "from_env"   # no override
    assert dotenv_lite.load_dotenv(str(tmp_path / "missing.env")) is False


def test_entry_points_read_dotenv_like_the_reference(tmp_path):
    """env.example sets EMBEDDING_MODEL_NAME / MILVUS_COLLECTION_NAME; importing the build module from a directory with
    a `.env` must see them (the reference: load_dotenv() at import, tools/build_database.py:11)"""
    (tmp_path / ".env").write_text("EMBEDDING_MODEL_NAME=shibing624/text2vec-base-chinese\nMILVUS_COLLECTION_NAME=icd10_collection\n")
    code = ("import os, sys; sys.path.insert(0, %r); import rag_project_icd10_amd.tools.build_database as b; "
            "print(os.environ.get('EMBEDDING_MODEL_NAME'), os.environ.get('MILVUS_COLLECTION_NAME'))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("EMBEDDING_MODEL_NAME", "MILVUS_COLLECTION_NAME")}
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["shibing624/text2vec-base-chinese", "icd10_collection"]


class _Emb:
    def __init__(self):
        self.batches = []

    def test_embedding(self, text="测试文本"):
        return {"success": True}

    def encode_query(self, text):
        return np.ones(8, np.float32)

    def encode_query_batch(self, texts, batch_size=256, to_device=False):
        self.batches.append(list(texts))
        return np.ones((len(texts), 8), np.float32)


class _Store:
    def __init__(self, fail_insert_at=None):
        self.rows, self.calls, self.fail_insert_at = [], [], fail_insert_at

    def test_connection(self):
        return {"connected": True}

    def clear_collection(self):
        self.calls.append("clear")
        self.rows = []
        return True

    def insert_records(self, records, embeddings):
        self.calls.append(("insert", len(records)))
        if self.fail_insert_at is not None and len(self.rows) >= self.fail_insert_at:
            return False
        assert len(records) == len(embeddings)
        self.rows.extend(records)
        return True

    def load_collection(self):
        self.calls.append("load")
        return True

    def get_collection_stats(self):
        return {"num_entities": len(self.rows)}

    def search(self, vec, top_k=10):
        self.calls.append(("search", top_k))
        return [{"code": r["code"]} for r in self.rows[:top_k]]


def _builder(monkeypatch, store):
    b = bd.DatabaseBuilder()
    emb = _Emb()

    def init():
        b.embedding_service, b.milvus_service = emb, store
    monkeypatch.setattr(b, "initialize_services", init)
    return b, emb


def test_build_full_database_orchestration(monkeypatch):
    csv = os.path.join(GOLDEN, "csv_slice.csv")
    store = _Store()
    b, emb = _builder(monkeypatch, store)
    assert b.build_full_database(csv, rebuild=True) is True
    n = len(store.rows)
    assert n == 273 and store.calls[0] == "clear"                               # --rebuild drops first (:304-307)
    assert [c for c in store.calls if isinstance(c, tuple) and c[0] == "insert"] == [("insert", 32)] * 8 + [("insert", 17)]   # < 1000 rows -> batches of 32
    assert store.calls[-1] == ("search", 5) and "load" in store.calls           # verify step: "急性胃肠炎", top 5 (:262-295)
    assert emb.batches[0][0] == store.rows[0]["semantic_text"]                  # the record's semantic_text is what gets embedded
    store2 = _Store()
    b2, _ = _builder(monkeypatch, store2)
    assert b2.build_full_database(csv, rebuild=False) is True and "clear" not in store2.calls   # incremental: no drop (:308-310)
    store3 = _Store(fail_insert_at=64)
    b3, _ = _builder(monkeypatch, store3)
    assert b3.build_full_database(csv) is False                                 # a failed insert aborts the build
    b4, _ = _builder(monkeypatch, _Store())
    assert b4.build_full_database(os.path.join(GOLDEN, "nope.csv")) is False    # any exception -> False (:335-337)


def test_cli_flags_and_exit_codes(monkeypatch, capsys):
    csv = os.path.join(GOLDEN, "csv_slice.csv")
    store = _Store()
    made = []

    class B(bd.DatabaseBuilder):
        def initialize_services(self):
            self.embedding_service, self.milvus_service = _Emb(), store
            made.append(self)
    monkeypatch.setattr(bd, "DatabaseBuilder", B)
    monkeypatch.setattr(sys, "argv", ["build_database.py", "--input", csv, "--rebuild"])
    assert bd.main() is True and store.calls[0] == "clear" and len(store.rows) == 273
    assert "数据库构建完成" in capsys.readouterr().out
    store.calls.clear()
    monkeypatch.setattr(sys, "argv", ["build_database.py", "--verify-only"])
    assert bd.main() is True
    assert not any(isinstance(c, tuple) and c[0] == "insert" for c in store.calls) and ("search", 5) in store.calls
    assert "数据库状态正常" in capsys.readouterr().out
    monkeypatch.setattr(sys, "argv", ["build_database.py", "--input", os.path.join(GOLDEN, "nope.csv")])
    assert bd.main() is False
    assert "数据库构建失败" in capsys.readouterr().out
    # the module's exit code follows main()'s return value (sys.exit(0 if main() else 1), :386-389)
    code = ("import sys; sys.path.insert(0, %r); sys.argv = ['x', '--input', %r]; "
            "import rag_project_icd10_amd.tools.build_database as b; "
            "b.DatabaseBuilder.initialize_services = lambda self: (_ for _ in ()).throw(RuntimeError('no gpu here')); "
            "sys.exit(0 if b.main() else 1)" % (ROOT, csv))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1
