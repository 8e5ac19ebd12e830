"""Request / response schemas of the /embed and /query surface (pydantic v2).

Field names, defaults and bounds follow the reference's models/icd_models.py: Candidate (:56-87, score
>= 0 at :71), DiagnosisMatch (:90-124), QueryRequest (:135-138, top_k in [1, 50], default 5),
QueryResponse (:141-158), EmbeddingRequest / EmbeddingResponse (:184-192), HealthCheckResponse
(:210-215), convert_numpy_types (:14-37).
"""
from __future__ import annotations

import dataclasses
from typing import Any, List, Optional

import numpy as np
from pydantic import BaseModel, ConfigDict, Field, field_serializer


def convert_numpy_types(obj):
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {k: convert_numpy_types(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [convert_numpy_types(v) for v in obj]
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        return convert_numpy_types(dataclasses.asdict(obj))
    if hasattr(obj, "__dict__"):
        return {k: convert_numpy_types(v) for k, v in obj.__dict__.items() if not k.startswith("_")}
    return obj


class Candidate(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    code: str = Field(..., description="ICD-10编码")
    title: str = Field(..., description="诊断名称")
    score: float = Field(..., description="相似度分数", ge=0.0)
    level: Optional[int] = Field(default=1, description="ICD层级级别")
    parent_code: Optional[str] = Field(default="", description="父级编码")
    enhanced_score: Optional[float] = Field(default=None, description="增强后的分数")
    original_score: Optional[float] = Field(default=None, description="原始相似度分数")
    similarity_factors: Optional[Any] = Field(default=None, description="相似度计算因子")

    @field_serializer("similarity_factors")
    def _ser_factors(self, value):
        return None if value is None else convert_numpy_types(value)


class DiagnosisMatch(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    diagnosis_text: str = Field(..., description="提取的诊断文本")
    candidates: List[Candidate] = Field(..., description="匹配的候选结果")
    match_confidence: float = Field(..., description="整体匹配置信度", ge=0.0, le=1.0)
    confidence_metrics: Optional[Any] = Field(default=None)
    confidence_factors: Optional[Any] = Field(default=None)
    confidence_level: Optional[str] = Field(default=None)

    @field_serializer("confidence_metrics", "confidence_factors")
    def _ser_conf(self, value):
        return None if value is None else convert_numpy_types(value)


class QueryRequest(BaseModel):
    text: str = Field(..., description="输入的诊断文本", min_length=1)
    top_k: int = Field(default=5, description="返回候选数量", ge=1, le=50)


class QueryResponse(BaseModel):
    model_config = ConfigDict(arbitrary_types_allowed=True)
    candidates: List[Candidate] = Field(..., description="候选结果列表")
    is_multi_diagnosis: bool = Field(default=False)
    extracted_diagnoses: List[str] = Field(default_factory=list)
    diagnosis_matches: List[DiagnosisMatch] = Field(default_factory=list)


class EmbeddingRequest(BaseModel):
    texts: List[str] = Field(..., description="要向量化的文本列表")


class EmbeddingResponse(BaseModel):
    embeddings: List[List[float]] = Field(..., description="向量列表")
    model: str = Field(..., description="使用的模型名称")


class HealthCheckResponse(BaseModel):
    status: str
    milvus_connected: bool
    embedding_model_loaded: bool
    total_records: int
