"""On-disk corpus format: the persistence that Milvus Lite's ./db/milvus_icd10.db file gives the
reference (services/milvus_service.py:70-82; SURVEY.md section 5 "Checkpoint / resume").

A store is a directory:
    manifest.json   {"format": 1, "collection": str, "dim": int, "count": int, "model": str|None}
    corpus.f32      row-major float32 [count][dim]      (exactly the layout icd_index_create takes)
    levels.i32      int32 [count]
    meta.jsonl      one JSON object per row: the 9 payload fields of the reference schema
                    (services/milvus_service.py:174-186): code, preferred_zh, has_complication, main_code,
                    secondary_code, level, parent_code, category_path, semantic_text
Rows are appended by `append()` (the reference's client.insert) and are durable once it returns. The manifest is
the commit record: it is rewritten (atomically) AFTER the three data files were extended, and carries their committed
lengths. A process that dies in between leaves surplus bytes behind the committed lengths; `_load()` and `append()`
truncate them (`_repair()`), so later rows never shift against their metadata. Files SHORTER than the manifest says
are corruption and raise.
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Any, Dict, List, Optional

import numpy as np

FORMAT_VERSION = 1
PAYLOAD_FIELDS = ("code", "preferred_zh", "has_complication", "main_code", "secondary_code", "level",
                  "parent_code", "category_path", "semantic_text")


class CorpusStore:
    def __init__(self, path: str, collection: str, dim: int):
        self.path = path
        self.collection = collection
        self.dim = int(dim)
        self.count = 0
        self.model: Optional[str] = None
        self.records: List[Dict[str, Any]] = []
        self._vectors: List[np.ndarray] = []   # chunks not yet concatenated
        self._matrix: Optional[np.ndarray] = None
        self._meta_bytes = 0   # committed length of meta.jsonl
        self.closed = False

    # ---- paths ------------------------------------------------------------------------------------
    def _dir(self) -> str:
        return os.path.join(self.path, self.collection)

    def _file(self, name: str) -> str:
        return os.path.join(self._dir(), name)

    def exists(self) -> bool:
        return os.path.exists(self._file("manifest.json"))

    # ---- open / create ----------------------------------------------------------------------------
    @classmethod
    def open(cls, path: str, collection: str, dim: int) -> "CorpusStore":
        st = cls(path, collection, dim)
        if st.exists():
            st._load()
        return st

    def create(self):
        os.makedirs(self._dir(), exist_ok=True)
        self.count = 0
        self._meta_bytes = 0
        self.records = []
        self._vectors = []
        self._matrix = np.zeros((0, self.dim), dtype=np.float32)
        for name in ("corpus.f32", "levels.i32", "meta.jsonl"):
            open(self._file(name), "wb").close()
        self._write_manifest()

    def drop(self):
        if os.path.isdir(self._dir()):
            shutil.rmtree(self._dir())
        self.count = 0
        self._meta_bytes = 0
        self.records = []
        self._vectors = []
        self._matrix = None

    def _write_manifest(self):
        tmp = self._file("manifest.json.tmp")
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump({"format": FORMAT_VERSION, "collection": self.collection, "dim": self.dim,
                       "count": self.count, "model": self.model, "meta_bytes": self._meta_bytes}, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self._file("manifest.json"))

    def _repair(self):
        """Cut the data files back to the committed lengths (bytes past them belong to an append that never committed);
        raise if a file is shorter than committed."""
        want = {"corpus.f32": self.count * self.dim * 4, "levels.i32": self.count * 4, "meta.jsonl": self._meta_bytes}
        for name, size in want.items():
            path = self._file(name)
            have = os.path.getsize(path) if os.path.exists(path) else -1
            if have < size:
                raise ValueError(f"{name} holds {have} bytes, the manifest committed {size}: the store is corrupt")
            if have > size:
                with open(path, "r+b") as f:
                    f.truncate(size)

    def _load(self):
        with open(self._file("manifest.json"), encoding="utf-8") as f:
            man = json.load(f)
        if man.get("format") != FORMAT_VERSION:
            raise ValueError(f"unsupported corpus store format {man.get('format')}")
        if int(man["dim"]) != self.dim:
            raise ValueError(f"store dimension {man['dim']} != expected {self.dim}")
        self.count = int(man["count"])
        self.model = man.get("model")
        if "meta_bytes" in man:
            self._meta_bytes = int(man["meta_bytes"])
        else:   # stores written before the field existed: the first `count` lines are the committed ones
            self._meta_bytes = 0
            with open(self._file("meta.jsonl"), "rb") as f:
                for i, line in enumerate(f):
                    if i >= self.count:
                        break
                    self._meta_bytes += len(line)
        self._repair()
        mat = np.fromfile(self._file("corpus.f32"), dtype=np.float32, count=self.count * self.dim)
        if mat.size != self.count * self.dim:
            raise ValueError("corpus.f32 is shorter than the manifest says")
        self._matrix = mat.reshape(self.count, self.dim)
        self._vectors = []
        self.records = []
        with open(self._file("meta.jsonl"), encoding="utf-8") as f:
            for i, line in enumerate(f):
                if i >= self.count:
                    break
                self.records.append(json.loads(line))
        if len(self.records) != self.count:
            raise ValueError("meta.jsonl is shorter than the manifest says")
        lv = np.fromfile(self._file("levels.i32"), dtype=np.int32, count=self.count)
        if lv.size != self.count or not np.array_equal(lv, self.levels()):
            raise ValueError("levels.i32 disagrees with meta.jsonl: the store is corrupt")

    # ---- rows ---------------------------------------------------------------------------------------
    def append(self, rows: List[Dict[str, Any]], vectors: np.ndarray):
        vectors = np.ascontiguousarray(vectors, dtype=np.float32).reshape(len(rows), self.dim)
        if not self.exists():
            self.create()
        self._repair()   # (an earlier append may have died before its manifest)
        meta = "".join(json.dumps({k: r.get(k) for k in PAYLOAD_FIELDS}, ensure_ascii=False) + "\n" for r in rows).encode("utf-8")
        with open(self._file("corpus.f32"), "ab") as f:
            vectors.tofile(f)
        with open(self._file("levels.i32"), "ab") as f:
            np.asarray([int(r.get("level", 1)) for r in rows], dtype=np.int32).tofile(f)
        with open(self._file("meta.jsonl"), "ab") as f:
            f.write(meta)
        self._meta_bytes += len(meta)
        self.records.extend(rows)
        self._vectors.append(vectors)
        self.count += len(rows)
        self._write_manifest()

    def matrix(self) -> np.ndarray:
        if self._vectors:
            parts = ([self._matrix] if self._matrix is not None and len(self._matrix) else []) + self._vectors
            self._matrix = np.concatenate(parts, axis=0) if parts else np.zeros((0, self.dim), np.float32)
            self._vectors = []
        if self._matrix is None:
            self._matrix = np.zeros((0, self.dim), dtype=np.float32)
        return self._matrix

    def levels(self) -> np.ndarray:
        """levels of the rows, from the metadata (levels.i32 is the same column as a flat file for external readers;
        `_load()` checks that the two agree)"""
        return np.asarray([int(r.get("level", 1)) for r in self.records], dtype=np.int32)

    def close(self):
        self.closed = True
