"""Vector store + search. Drop-in for the reference's services/milvus_service.py with the Milvus
engine replaced by an HBM-resident index searched by hand-written HIP kernels (libicdsearch.so).

Same public surface and return shapes: search (:271-320), insert_records (:208-269),
get_collection_stats (:322-341), load_collection (:343-357), clear_collection (:359-371),
test_connection (:373-408), release_collection (:410-434), get_collection_load_state (:436-458),
disconnect (:460-498), get_memory_usage (:500-522), health_check (:524-549),
_calculate_level_weight (:550-558); attributes .config .collection_name .embedding_service
.dimension .client (read by main.py and tools/build_database.py).

`MILVUS_DB_PATH` names a directory holding the on-disk corpus (corpus_store.py) instead of a Milvus
Lite file. `MILVUS_MODE=remote` (a gRPC client of a Milvus server, :84-111) is out of scope and raises.
There is no CPU search path: without libicdsearch.so / an MI355X, loading the collection fails and
`search` returns [] exactly as the reference does on engine errors (:318-320).

Additive: `search_batch` (many queries per call, numpy or device tensors).
"""
from __future__ import annotations

import datetime
import logging
import os
from typing import Any, Dict, List, Optional

import numpy as np

from ..corpus_store import CorpusStore

logger = logging.getLogger(__name__)

_OUTPUT_FIELDS = ("code", "preferred_zh", "has_complication", "main_code", "secondary_code", "level",
                  "parent_code", "category_path", "semantic_text")


class MilvusService:
    def __init__(self, embedding_service=None):
        self.config = self._load_config()
        self.collection_name = self.config.get("milvus", {}).get("collection_name", "icd10")
        self.embedding_service = embedding_service
        self.dimension = self._get_vector_dimension()
        self.client: Optional[CorpusStore] = None
        self._index = None          # rag_project_icd10_amd._native.IcdIndex
        self._index_rows = 0
        self._row_tags = None
        self._connect()
        self._setup_collection()

    # ---- configuration (reference :21-55) -------------------------------------------------------------
    def _load_config(self) -> Dict[str, Any]:
        return {"milvus": {
            "mode": os.getenv("MILVUS_MODE", "local"),
            "host": os.getenv("MILVUS_HOST", "localhost"),
            "port": int(os.getenv("MILVUS_PORT", "19530")),
            "username": os.getenv("MILVUS_USERNAME", ""),
            "password": os.getenv("MILVUS_PASSWORD", ""),
            "db_name": os.getenv("MILVUS_DB_NAME", "default"),
            "db_path": os.getenv("MILVUS_DB_PATH", "./db/milvus_icd10.db"),
            "collection_name": os.getenv("MILVUS_COLLECTION_NAME", "icd10"),
            "index_type": "FLAT",
            "metric_type": "IP",
            "secure": os.getenv("MILVUS_SECURE", "false").lower() == "true",
            # knobs of this build
            "gpu_device": int(os.getenv("ICD_GPU_DEVICE", os.getenv("LOCAL_RANK", "0"))),
            "max_batch": int(os.getenv("ICD_GPU_MAX_BATCH", "16384")),
            "max_k": int(os.getenv("ICD_GPU_MAX_K", "100")),
        }}

    def _get_vector_dimension(self) -> int:
        if self.embedding_service:
            try:
                return len(self.embedding_service.encode_query("测试文本"))
            except Exception as exc:
                logger.warning("无法从嵌入服务获取维度: %s", exc)
        return 1024

    def _connect(self):
        cfg = self.config.get("milvus", {})
        mode = cfg.get("mode", "local")
        try:
            if mode == "local":
                path = cfg.get("db_path", "./db/milvus_icd10.db")
                os.makedirs(path, exist_ok=True)
                self.client = CorpusStore.open(path, self.collection_name, self.dimension)
            elif mode == "remote":
                raise ValueError("MILVUS_MODE=remote (client of a Milvus server) is not part of this build; use 'local'")
            else:
                raise ValueError(f"不支持的Milvus模式: {mode}，请使用 'local' 或 'remote'")
        except Exception as exc:
            logger.error("Milvus连接失败 (模式: %s): %s", mode, exc)
            raise

    def _setup_collection(self):
        if not self.client.exists():
            self._create_collection()
        self._load_collection_to_memory()

    def _create_collection(self):
        self.client.create()

    # ---- HBM residency -----------------------------------------------------------------------------------
    def _load_collection_to_memory(self):
        """Upload the corpus to HBM (the reference's load_collection, :137-161). An empty collection
        counts as loaded. Raises if the native library / GPU is unavailable."""
        n = self.client.count
        if n == 0:
            self._drop_index()
            self._loaded = True
            return
        if self._index is not None and self._index_rows == n:
            self._loaded = True
            return
        from .._native import IcdIndex
        self._drop_index()
        cfg = self.config["milvus"]
        self._index = IcdIndex(self.client.matrix(), self.client.levels(), device=cfg["gpu_device"],
                               max_nq=cfg["max_batch"], max_k=cfg["max_k"])
        self._index_rows = n
        self._loaded = True

    def _drop_index(self):
        if self._index is not None:
            self._index.close()
        self._index = None
        self._index_rows = 0
        self._row_tags = None
        self._loaded = False

    def supports_device_rescoring(self) -> bool:
        """True when search_batch returns device tensors that icd_hier_rescore can take (an index in HBM on a GPU)"""
        try:
            import torch
            return torch.cuda.is_available() and self._ready_index() is not None
        except Exception:
            return False

    def row_tags(self):
        """uint8 [n] on the index's GPU: what the device-side hierarchical rescoring needs to know about each row's code
        (HierarchicalSimilarityService.row_tag); built on first use after a load."""
        index = self._ready_index()
        if index is None:
            raise RuntimeError(f"collection {self.collection_name} is empty or missing")
        if getattr(self, "_row_tags", None) is None or self._row_tags.numel() != self.client.count:
            import torch
            from .hierarchical_similarity_service import HierarchicalSimilarityService as H
            tags = np.fromiter((H.row_tag(r.get("code") or "") for r in self.client.records), dtype=np.uint8, count=self.client.count)
            self._row_tags = torch.from_numpy(tags).to(torch.device("cuda", self.config["milvus"]["gpu_device"]))
        return self._row_tags

    # ---- writes -------------------------------------------------------------------------------------------
    def insert_records(self, records: List[Dict[str, Any]], embeddings: List[np.ndarray]) -> bool:
        if len(records) != len(embeddings):
            raise ValueError("记录数量与向量数量不匹配")
        try:
            rows, vecs = [], []
            for i, rec in enumerate(records):
                secondary = rec.get("secondary_code")
                main = rec.get("main_code")
                vecs.append(embeddings[i].tolist())  # a plain list here fails like the reference (:231)
                rows.append({
                    "code": rec["code"],
                    "preferred_zh": rec.get("preferred_zh", ""),
                    "has_complication": rec.get("has_complication", False),
                    "main_code": "" if main is None else main,
                    "secondary_code": "" if secondary is None else secondary,
                    "level": rec.get("level", 1),
                    "parent_code": rec.get("parent_code", ""),
                    "category_path": rec.get("category_path", ""),
                    "semantic_text": rec.get("semantic_text", ""),
                })
            mat = np.asarray(vecs, dtype=np.float32)
            if mat.ndim != 2 or mat.shape[1] != self.dimension:
                raise ValueError(f"vector dimension {mat.shape} != {self.dimension}")
            self.client.append(rows, mat)
            self._loaded = self._index is not None and self._index_rows == self.client.count
            return True
        except Exception as exc:
            logger.error("插入记录失败: %s", exc)
            return False

    # ---- search ---------------------------------------------------------------------------------------------
    def _ready_index(self):
        if self.client is None or not self.client.exists():
            return None
        if self._index is None or self._index_rows != self.client.count:
            self._load_collection_to_memory()
        return self._index

    def search(self, query_vector: np.ndarray, top_k: int = 10) -> List[Dict[str, Any]]:
        try:
            if self.client is None or not self.client.exists():
                logger.error("集合 %s 不存在", self.collection_name)
                return []
            index = self._ready_index()
            if index is None:
                return []
            # (the reference sends data=[query_vector.tolist()], :282: a plain list has no .tolist and lands in the except below,
            #  an array's values arrive as float32 either way - converting through a Python list cost 35 us of a 130-us call)
            query_vector.tolist  # noqa: B018
            q = np.asarray(query_vector, dtype=np.float32).reshape(1, -1)
            adj, raw, ids, levels = index.search_reweighted(q, int(top_k))
            return self._hits_to_dicts(adj[0], raw[0], ids[0])
        except Exception as exc:
            logger.error("搜索失败: %s", exc)
            return []

    def _hits_to_dicts(self, adj, raw, ids) -> List[Dict[str, Any]]:
        out = []
        recs = self.client.records
        for a, r, i in zip(adj, raw, ids):
            i = int(i)
            if i < 0:
                continue
            hit = recs[i]
            out.append({
                "code": hit.get("code"),
                "title": hit.get("preferred_zh"),
                "score": float(a),
                "original_score": float(r),
                "metadata": {
                    "has_complication": hit.get("has_complication", False),
                    "main_code": hit.get("main_code", ""),
                    "secondary_code": hit.get("secondary_code", ""),
                    "level": hit.get("level", 1),
                    "parent_code": hit.get("parent_code", ""),
                    "category_path": hit.get("category_path", ""),
                    "semantic_text": hit.get("semantic_text", ""),
                },
            })
        return out

    def search_batch(self, query_vectors, top_k: int = 10, as_dicts: bool = False):
        """Additive: many queries in one call. query_vectors: [nq, dim] numpy array or torch CUDA
        tensor. Returns (adjusted f64, raw f32, ids i64, levels i32), each [nq, top_k], in the order
        `search` returns hits; or, with as_dicts=True, a list of `search`-shaped hit lists."""
        index = self._ready_index()
        if index is None:
            raise RuntimeError(f"collection {self.collection_name} is empty or missing")
        # (large batches on a corpus of tight families of near-identical rows - ICD sibling codes - are handled inside the
        #  library: a second coarse pass over the queries the first could not certify, and from the next large batch on the
        #  wider partition right away; include/icd_search.h icd_stats.last_second_pass / wide_mode)
        adj, raw, ids, levels = index.search_reweighted(query_vectors, int(top_k))
        if not as_dicts:
            return adj, raw, ids, levels
        if hasattr(adj, "cpu"):
            adj, raw, ids = adj.cpu().numpy(), raw.cpu().numpy(), ids.cpu().numpy()
        return [self._hits_to_dicts(adj[q], raw[q], ids[q]) for q in range(len(ids))]

    # ---- admin (same keys as the reference) -------------------------------------------------------------------------
    def get_collection_stats(self) -> Dict[str, Any]:
        try:
            exists = self.client is not None and self.client.exists()
            return {"collection_name": self.collection_name, "exists": exists, "dimension": self.dimension,
                    "num_entities": self.client.count if exists else 0}
        except Exception as exc:
            return {"error": str(exc)}

    def load_collection(self) -> bool:
        try:
            if self.client is None or not self.client.exists():
                logger.error("集合 %s 不存在", self.collection_name)
                return False
            self._load_collection_to_memory()
            return True
        except Exception as exc:
            logger.error("加载集合失败: %s", exc)
            return False

    def clear_collection(self) -> bool:
        try:
            self._drop_index()
            self.client.drop()
            self._setup_collection()
            return True
        except Exception as exc:
            logger.error("清空集合失败: %s", exc)
            return False

    def test_connection(self) -> Dict[str, Any]:
        mode = self.config.get("milvus", {}).get("mode", "local")
        try:
            if self.client is None:
                raise RuntimeError("客户端未连接")
            return {"connected": True, "mode": mode, "collection_stats": self.get_collection_stats(),
                    "client_type": "IcdIndex(MI355X)",
                    "local_info": {"db_path": self.config.get("milvus", {}).get("db_path")}}
        except Exception as exc:
            return {"connected": False, "error": str(exc), "mode": mode}

    def release_collection(self) -> Dict[str, Any]:
        try:
            if not self.client:
                return {"success": False, "message": "客户端未连接"}
            if not self.client.exists():
                return {"success": False, "message": f"集合 {self.collection_name} 不存在"}
            self._drop_index()
            return {"success": True, "message": f"集合 {self.collection_name} 内存已释放",
                    "collection_name": self.collection_name}
        except Exception as exc:
            return {"success": False, "message": f"释放集合内存失败: {exc}"}

    def get_collection_load_state(self) -> Dict[str, Any]:
        try:
            if not self.client:
                return {"loaded": False, "message": "客户端未连接"}
            if not self.client.exists():
                return {"loaded": False, "message": f"集合 {self.collection_name} 不存在"}
            loaded = bool(getattr(self, "_loaded", False)) and (self._index is not None or self.client.count == 0)
            return {"loaded": loaded, "state": "Loaded" if loaded else "NotLoad", "collection_name": self.collection_name}
        except Exception as exc:
            return {"loaded": False, "message": f"获取集合加载状态失败: {exc}"}

    def disconnect(self) -> Dict[str, Any]:
        try:
            if not self.client:
                return {"success": True, "message": "客户端已经断开"}
            release_result = self.release_collection()
            self.client.close()
            self.client = None
            return {"success": True, "message": "Milvus连接已断开，资源已清理", "release_result": release_result}
        except Exception as exc:
            return {"success": False, "message": f"断开Milvus连接失败: {exc}"}

    def get_memory_usage(self) -> Dict[str, Any]:
        try:
            if not self.client:
                return {"memory_usage": 0, "message": "客户端未连接"}
            if not self.client.exists():
                return {"memory_usage": 0, "message": f"集合 {self.collection_name} 不存在"}
            stats = self.get_collection_stats()
            state = self.get_collection_load_state()
            out = {
                "collection_name": self.collection_name,
                "loaded": state.get("loaded", False),
                "load_state": state.get("state", "Unknown"),
                "num_entities": stats.get("num_entities", 0),
                "estimated_memory_mb": stats.get("num_entities", 0) * self.dimension * 4 / (1024 * 1024),
                "message": "内存使用为估算值（基于向量维度和实体数量）",
            }
            if self._index is not None:
                st = self._index.stats()
                out["hbm_bytes"] = st["bytes_corpus_f32"] + st["bytes_corpus_f16"] + st["bytes_workspace"]
            return out
        except Exception as exc:
            return {"memory_usage": 0, "message": f"获取内存使用情况失败: {exc}"}

    def health_check(self) -> Dict[str, Any]:
        try:
            conn = self.test_connection()
            state = self.get_collection_load_state()
            return {"healthy": conn.get("connected", False) and state.get("loaded", False), "connection": conn,
                    "load_state": state, "memory_usage": self.get_memory_usage(),
                    "timestamp": datetime.datetime.now().isoformat()}
        except Exception as exc:
            return {"healthy": False, "error": str(exc), "timestamp": datetime.datetime.now().isoformat()}

    def _calculate_level_weight(self, level: int) -> float:
        return {1: 1.2, 2: 1.0, 3: 0.8}.get(level, 1.0)
