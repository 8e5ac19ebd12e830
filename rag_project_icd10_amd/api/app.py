"""FastAPI surface: POST /embed, POST /query, GET /health, GET /stats, GET /.

Mirrors the reference's main.py for these routes: lifespan-owned service globals (:25-105), `/`
(:250-258), `/health` (:261-289), `/query` (:292-363: candidates of all matches merged, sorted by score,
cut to top_k; 503 when services are missing, 500 with a `detail` string on any exception), `/embed`
(:505-530), `/stats` (:574-599). The LLM, NER, standardisation and resource routes are out of scope.

    uvicorn rag_project_icd10_amd.api.app:app --host 0.0.0.0 --port 8005
"""
from __future__ import annotations

import logging
import os
from contextlib import asynccontextmanager

from fastapi import FastAPI, HTTPException

from ..dotenv_lite import load_dotenv
from .icd_models import (DiagnosisMatch, EmbeddingRequest, EmbeddingResponse, HealthCheckResponse,
                         QueryRequest, QueryResponse, convert_numpy_types)

load_dotenv()   # main.py:11 of the reference; existing environment variables win

logger = logging.getLogger(__name__)

embedding_service = None
milvus_service = None
multi_diagnosis_service = None


def install_services(embedding, milvus, multi=None):
    """Wire service instances (used by the lifespan and by tests)."""
    global embedding_service, milvus_service, multi_diagnosis_service
    embedding_service, milvus_service = embedding, milvus
    if multi is None and embedding is not None and milvus is not None:
        from ..services.multi_diagnosis_service import MultiDiagnosisService
        multi = MultiDiagnosisService(embedding, milvus)
    multi_diagnosis_service = multi


@asynccontextmanager
async def lifespan(app: FastAPI):
    if embedding_service is None:
        from ..services.embedding_service import EmbeddingService
        from ..services.milvus_service import MilvusService
        emb = EmbeddingService()
        mil = MilvusService(emb)
        # like the reference's MultiDiagnosisService (services/multi_diagnosis_service.py:28,44-47): an NER service - its classifier when the
        # checkpoint resolves, its rules otherwise - and with it the ENHANCED text mode (entities fused with semantic boundaries);
        # ICD_QUERY_NER=0: neither (delimiter extraction, the whole request on the device)
        ner = None
        if os.getenv("ICD_QUERY_NER", "1") != "0":
            try:
                from ..services.medical_ner_service import MedicalNERService
                ner = MedicalNERService()
            except Exception as exc:
                logger.error("no NER service (%s): /query extracts by delimiters", exc)
        from ..services.multi_diagnosis_service import MultiDiagnosisService
        install_services(emb, mil, MultiDiagnosisService(emb, mil, ner_service=ner))
    try:
        yield
    finally:
        if milvus_service is not None:
            try:
                milvus_service.disconnect()
            except Exception as exc:
                logger.warning("disconnect failed: %s", exc)


app = FastAPI(title="ICD-10 诊断标准化API", description="基于RAG的ICD-10诊断内容标准化系统 (MI355X)",
              version="1.0.0", lifespan=lifespan)


@app.get("/")
async def root():
    return {"message": "ICD-10 诊断标准化API", "version": "1.0.0", "docs": "/docs", "health": "/health"}


@app.get("/health", response_model=HealthCheckResponse)
async def health_check():
    try:
        loaded = bool(embedding_service and embedding_service.get_model_info().get("loaded", False))
        connected, total = False, 0
        if milvus_service:
            connected = milvus_service.test_connection().get("connected", False)
            if connected:
                total = milvus_service.get_collection_stats().get("num_entities", 0)
        return HealthCheckResponse(status="healthy" if (loaded and connected) else "unhealthy",
                                   milvus_connected=connected, embedding_model_loaded=loaded, total_records=total)
    except Exception as exc:
        raise HTTPException(status_code=500, detail=f"健康检查失败: {exc}")


@app.post("/query", response_model=QueryResponse)
async def query_similar(request: QueryRequest):
    try:
        if not embedding_service or not milvus_service or not multi_diagnosis_service:
            raise HTTPException(status_code=503, detail="服务未就绪")
        result = multi_diagnosis_service.match_multiple_diagnoses(text=request.text, top_k=request.top_k)
        candidates, matches = [], []
        for m in result["matches"]:
            candidates.extend(m.candidates)
            matches.append(DiagnosisMatch(diagnosis_text=m.diagnosis_text, candidates=m.candidates,
                                          match_confidence=m.match_confidence))
        candidates.sort(key=lambda c: c.score, reverse=True)
        response = QueryResponse(candidates=candidates[:request.top_k],
                                 is_multi_diagnosis=len(result["extracted_diagnoses"]) > 1,
                                 extracted_diagnoses=result["extracted_diagnoses"], diagnosis_matches=matches)
        try:
            response = QueryResponse(**convert_numpy_types(response.model_dump()))
        except Exception as exc:
            logger.warning("numpy conversion failed: %s", exc)
        return response
    except Exception as exc:
        # like the reference (:361-363) every failure, the 503 above included, surfaces as a 500
        raise HTTPException(status_code=500, detail=f"查询失败: {exc}")


@app.post("/embed", response_model=EmbeddingResponse)
async def embed_texts(request: EmbeddingRequest):
    try:
        if not embedding_service:
            raise HTTPException(status_code=503, detail="向量化服务未就绪")
        embeddings = embedding_service.encode_batch(request.texts, show_progress=False)
        name = embedding_service.get_model_info().get("model_name", "unknown")
        return EmbeddingResponse(embeddings=embeddings, model=name)
    except Exception as exc:
        raise HTTPException(status_code=500, detail=f"向量化失败: {exc}")


@app.get("/stats")
async def get_stats():
    try:
        stats = {}
        if milvus_service:
            stats["milvus"] = milvus_service.get_collection_stats()
        if embedding_service:
            stats["embedding"] = embedding_service.get_model_info()
        return stats
    except Exception as exc:
        raise HTTPException(status_code=500, detail=f"获取统计信息失败: {exc}")
