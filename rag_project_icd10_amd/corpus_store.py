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
the commit record: it is rewritten (atomically) AFTER the three data files were extended AND fsynced, and carries their
committed lengths. A process that dies in between leaves surplus bytes behind the committed lengths; `_load()` and
`append()` truncate them (`_repair()`), so later rows never shift against their metadata. Files SHORTER than the
manifest says are corruption and raise.

Several processes may hold handles of one store (`build_database` next to the API): every load, repair and append runs
under an exclusive flock on `<store>/.lock`, and `append()` first compares the manifest on disk with what this handle
last saw - a handle that another process has appended behind RELOADS before it appends (it never truncates rows that
were committed by someone else).
"""
from __future__ import annotations

import contextlib
import fcntl
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

    @contextlib.contextmanager
    def _locked(self):
        """exclusive inter-process lock of the store directory (flock: released by the kernel if the holder dies)"""
        os.makedirs(self._dir(), exist_ok=True)
        fd = os.open(self._file(".lock"), os.O_CREAT | os.O_RDWR, 0o644)
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
            yield
        finally:
            fcntl.flock(fd, fcntl.LOCK_UN)
            os.close(fd)

    def _fsync_dir(self):
        fd = os.open(self._dir(), os.O_RDONLY)
        try:
            os.fsync(fd)
        finally:
            os.close(fd)

    # ---- open / create ----------------------------------------------------------------------------
    @classmethod
    def open(cls, path: str, collection: str, dim: int) -> "CorpusStore":
        st = cls(path, collection, dim)
        if st.exists():
            with st._locked():
                st._load()
        return st

    def create(self):
        with self._locked():
            self._create()

    def _create(self):
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
        self._fsync_dir()   # the rename itself

    def _disk_commit(self):
        """(count, meta_bytes) of the manifest on disk, or None"""
        try:
            with open(self._file("manifest.json"), encoding="utf-8") as f:
                man = json.load(f)
            return int(man["count"]), (int(man["meta_bytes"]) if "meta_bytes" in man else None)
        except FileNotFoundError:
            return None

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
        meta = "".join(json.dumps({k: r.get(k) for k in PAYLOAD_FIELDS}, ensure_ascii=False) + "\n" for r in rows).encode("utf-8")
        with self._locked():
            disk = self._disk_commit()
            if disk is None:
                self._create()
            elif disk != (self.count, self._meta_bytes):
                self._load()     # another handle committed rows since this one looked: take them in, never cut them off
            else:
                self._repair()   # (an earlier append may have died before its manifest: surplus bytes of nobody's)

            def extend(name, write):
                with open(self._file(name), "ab") as f:
                    write(f)
                    f.flush()
                    os.fsync(f.fileno())   # durable BEFORE the manifest that commits it

            extend("corpus.f32", vectors.tofile)
            extend("levels.i32", np.asarray([int(r.get("level", 1)) for r in rows], dtype=np.int32).tofile)
            extend("meta.jsonl", lambda f: f.write(meta))
            self._fsync_dir()
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

    def code_title_columns(self):
        """(codes, titles): the two payload fields a Candidate is made of, as plain lists by row - rebuilt when rows were added
        (the batched request path builds 10 000 Candidates per 1 000 strings: list indexing, not two dict lookups per object)"""
        cached = getattr(self, "_code_title", None)
        if cached is None or cached[0] != len(self.records):
            codes = [r.get("code", "") for r in self.records]
            titles = [r.get("preferred_zh", "") for r in self.records]
            self._code_title = cached = (len(self.records), codes, titles)
        return cached[1], cached[2]

    def levels(self) -> np.ndarray:
        """levels of the rows, from the metadata (levels.i32 is the same column as a flat file for external readers;
        `_load()` checks that the two agree)"""
        return np.asarray([int(r.get("level", 1)) for r in self.records], dtype=np.int32)

    def close(self):
        self.closed = True
