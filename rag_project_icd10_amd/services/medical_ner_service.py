"""Medical named-entity recognition: a BERT token classifier on ROCm for whole batches, or regular-expression rules.

Follows the reference's services/medical_ner_service.py: same constructor arguments and environment
variables (MEDICAL_NER_MODEL, USE_MEDICAL_NER_MODEL, MEDICAL_NER_MIN_CONFIDENCE; :18-65), same label -> type table,
`extract_medical_entities(text, filter_drugs=True)` with model first, rules on failure, then the diagnosis filter
(:144-176), the conversion of the classifier's entity groups (:177-229), the rule patterns with their validity and
confidence rules (:102-142, :230-320), the overlap de-duplication (:322-351), `identify_diagnosis_keywords`,
`get_model_info`, `get_entity_summary`, `get_filter_stats` (:353-464).

What the reference delegates to transformers' `pipeline("ner", aggregation_strategy="simple")` (:71-92; transformers is
an un-vendored dependency, `requirements.txt`) is restated in `_TokenClassifier`: tokenise with character offsets,
softmax of the logits in float32, one label per token (argmax), adjacent tokens of the same tag merged unless the
later one is a B- token, group score = mean of the token scores, group word = the tokens joined, "O" groups dropped.
Pinned against the pipeline itself on a seeded model (tests/golden/make_ner_golden.py, tests/test_ner_cpu.py).
Differences by design: `extract_medical_entities_batch` runs ONE padded forward on the GPU for many strings (the
reference's /query issues 1 + 3 D single-string forwards per request); weights are resolved locally only - if the
checkpoint is not available the service switches to the rules. DELIBERATE DEVIATION: the reference's failed load sets
`use_model = False` WITHOUT building the rule patterns (:93-101), so its rules path then raises AttributeError on every
call and /query degrades each diagnosis to "no entities"; here a failed load also calls `_init_fallback_patterns()`, so
the rules really run (an offline deployment gets entities instead of none) - unless synthetic weights are explicitly allowed (ICD_NER_ALLOW_SYNTHETIC=1: a seeded random-init BERT-base token
classifier of the same shape, for throughput measurements; its entities mean nothing).
"""
from __future__ import annotations

import logging
import os
import re
from typing import Any, Dict, List, Sequence

import numpy as np

from .. import graph_capture
from .diagnosis_entity_filter import DiagnosisEntityFilter

logger = logging.getLogger(__name__)

DEFAULT_NER_MODEL = "lixin12345/chinese-medical-ner"

ENTITY_TYPE_MAPPING = {
    "DiseaseNameOrComprehensiveCertificate": "disease", "Symptom": "symptom", "BodyParts": "anatomy",
    "OrganOrCellDamage": "pathology", "Drug": "drug", "TreatmentOrPreventionProcedures": "treatment",
    "TreatmentEquipment": "equipment", "InspectionProcedure": "inspection", "MedicalTestingItems": "lab_indicator",
    "Department": "department", "Sign": "sign", "InjuryOrPoisoning": "injury", "Microbiology": "microbiology",
    "MedicalProcedures": "procedure", "InspectEquipment": "inspect_equipment",
}

_X = r"[^，。；\s]"   # one character of an entity body
MEDICAL_PATTERNS = {
    "disease": [
        rf"(?:急性|慢性|原发性|继发性|复发性|亚急性)?{_X}{{2,12}}(?:病|症|炎|癌|瘤|综合征)",
        rf"(?:急性|慢性)?{_X}{{2,8}}(?:感染|中毒|损伤|破裂|梗死|出血)",
        rf"(?:I|II|III|IV|V)+型{_X}{{2,8}}(?:病|症)",
        rf"{_X}{{2,8}}(?:功能不全|功能障碍|衰竭)",
    ],
    "symptom": [
        rf"(?:反复|持续|间歇性|突发性)?{_X}{{2,6}}(?:痛|疼|热|胀|肿|晕|麻|痒)",
        rf"(?:大量|少量|血性|脓性)?{_X}{{2,6}}(?:出血|分泌|呕吐|腹泻)",
        rf"{_X}{{2,6}}(?:不适|异常|增大|缩小|肥厚)",
        rf"(?:阵发性|持续性)?{_X}{{2,6}}(?:咳嗽|气促|心悸|失眠)",
    ],
    "anatomy": [
        rf"(?:左|右|双侧|上|下|前|后)?(?:心|肝|肺|肾|胃|肠|脑|骨|脊柱){_X}{{0,6}}",
        rf"(?:左|右|双侧)?(?:乳腺|甲状腺|前列腺|子宫|卵巢){_X}{{0,4}}",
        rf"(?:颈|胸|腰|骶|尾)椎{_X}{{0,4}}",
        rf"(?:主|冠状|肺|肾)动脉{_X}{{0,4}}",
    ],
}
STOP_WORDS = frozenset((
    "待查", "考虑", "疑似", "排除", "？", "?", "诊断为", "患者", "病人", "检查", "发现", "显示", "提示", "建议", "需要",
    "进一步", "复查", "治疗", "用药", "服用", "注射", "输液", "手术", "康复"))
MEANINGLESS_PHRASES = frozenset(("不详", "不明", "不清", "未明确", "待定", "观察", "随访"))
# rule confidence: +0.2 when the text carries one of the type's marker characters (:290-318)
_TYPE_MARKERS = {"disease": ("病", "症", "炎", "癌", "瘤"), "symptom": ("痛", "热", "胀", "肿", "出血"),
                 "anatomy": ("心", "肝", "肺", "肾", "脑")}


class _CharOffsetTokenizer:
    """Stand-in when no tokenizer files are available: one token per non-space character with its character offsets
    (what bert-base-chinese does for CJK text), deterministic ids."""
    cls_id, sep_id, pad_id, unk_id = 101, 102, 0, 100

    def __init__(self, vocab_size: int, max_len: int = 512):
        self.vocab_size, self.max_len = vocab_size, max_len

    def encode(self, text: str):
        kept = [(i, ch.lower()) for i, ch in enumerate(text) if not ch.isspace()][: self.max_len - 2]
        span = self.vocab_size - 1000
        ids = [self.cls_id] + [1000 + ord(ch) % span for _, ch in kept] + [self.sep_id]
        offsets = [(0, 0)] + [(i, i + 1) for i, _ in kept] + [(0, 0)]
        tokens = ["[CLS]"] + [ch for _, ch in kept] + ["[SEP]"]
        return ids, offsets, tokens, [1] + [0] * len(kept) + [1]

    @staticmethod
    def join(tokens: Sequence[str]) -> str:
        return " ".join(tokens).replace(" ##", "").strip()


class _TokenClassifier:
    """A token-classification forward for a batch of strings + transformers' 'simple' aggregation, restated.

    model: a module whose forward(input_ids=, attention_mask=) returns an object with `.logits` [B, T, L];
    tokenizer: a transformers fast tokenizer (offset mapping) or a _CharOffsetTokenizer; id2label: {int: str}."""

    PACK_TOKENS = 65536

    def __init__(self, model, tokenizer, id2label: Dict[int, str], device: str = "cpu", max_batch: int = 256):
        import torch
        self.torch = torch
        self.model = model.to(device).eval()
        self.tokenizer = tokenizer
        self.id2label = {int(k): v for k, v in id2label.items()}
        self.device = device
        self.max_batch = max_batch
        # BERT token classifiers (`.bert` + `.classifier`): batches above 32 strings run the encoder over packed tokens
        # (embedding_service._PackedBert: no GEMM work on pad tokens, one GEMM per Linear over the whole batch) and the
        # classifier head over the same packed rows. Anything else, and small batches, keep the padded forward.
        self._packed = None
        try:
            from .embedding_service import _PackedBert
            bert = getattr(self.model, "bert", None) or getattr(self.model, "roberta", None)   # (BertFor... / (XLM)RobertaForTokenClassification)
            if (os.getenv("ICD_NER_PACKED", "1") == "1" and bert is not None and hasattr(self.model, "classifier")
                    and _PackedBert.supported(bert)):
                self._packed = _PackedBert(bert)
        except Exception as exc:   # pragma: no cover - an optimisation, never fatal
            logger.warning("packed NER forward unavailable (%s): padded batches", exc)
        # one request's one to a few strings (<= 32 strings, <= 256 tokens in all): the hand-written small-input forward, ONE graph
        # launch for the encoder (csrc/encoder_small.hpp through icd_encoder_encode), the classifier head over its packed last
        # hidden state. fp32 BERT-base shapes on a GPU; ICD_NER_SMALL=0 keeps the replayed graph of the framework's forward.
        self._small = None
        try:
            bert = getattr(self.model, "bert", None) or getattr(self.model, "roberta", None)
            if (os.getenv("ICD_NER_SMALL", "1") == "1" and str(device).startswith("cuda") and bert is not None
                    and hasattr(self.model, "classifier")):
                from .. import _native
                if _native.SmallEncoder.supported(bert):
                    self._small = _native.SmallEncoder(bert)
        except Exception as exc:   # pragma: no cover - an optimisation, never fatal
            logger.info("small-input NER forward unavailable (%s): the framework's forward", exc)

    # -- tokenisation: ids, character offsets, token strings, special-token mask ---------------------------------
    def _encode(self, text: str):
        tk = self.tokenizer
        if isinstance(tk, _CharOffsetTokenizer):
            return tk.encode(text)
        limit = getattr(tk, "model_max_length", 0)
        enc = tk(text, return_offsets_mapping=True, return_special_tokens_mask=True,
                 truncation=bool(limit and 0 < limit < 10 ** 9))
        ids = list(enc["input_ids"])
        tokens = tk.convert_ids_to_tokens(ids)
        unk = tk.unk_token_id
        offsets = [tuple(o) for o in enc["offset_mapping"]]
        tokens = [text[offsets[i][0]:offsets[i][1]] if ids[i] == unk else tokens[i] for i in range(len(ids))]
        return ids, offsets, tokens, list(enc["special_tokens_mask"])

    def _encode_all(self, texts: Sequence[str]):
        """_encode for many strings: ONE call into the fast tokenizer (its Rust side encodes the batch in parallel; one
        Python call per string was 0.15 ms of 0.38 ms per string on the host). No padding is requested, so every string
        gets what the single call gives it."""
        tk = self.tokenizer
        if isinstance(tk, _CharOffsetTokenizer) or len(texts) < 2:
            return [self._encode(t) for t in texts]
        limit = getattr(tk, "model_max_length", 0)
        enc = tk(list(texts), return_offsets_mapping=True, return_special_tokens_mask=True,
                 truncation=bool(limit and 0 < limit < 10 ** 9))
        unk = tk.unk_token_id
        out = []
        for i, text in enumerate(texts):
            ids = list(enc["input_ids"][i])
            offsets = [tuple(o) for o in enc["offset_mapping"][i]]
            tokens = tk.convert_ids_to_tokens(ids)
            if unk in ids:
                tokens = [text[offsets[t][0]:offsets[t][1]] if ids[t] == unk else tokens[t] for t in range(len(ids))]
            out.append((ids, offsets, tokens, list(enc["special_tokens_mask"][i])))
        return out

    def _join(self, tokens: Sequence[str]) -> str:
        tk = self.tokenizer
        return tk.join(tokens) if isinstance(tk, _CharOffsetTokenizer) else tk.convert_tokens_to_string(list(tokens))

    # -- small batches replay a captured HIP graph (a request's one to a few strings: ~150 tiny kernels whose launch overhead,
    # not their arithmetic, sets the latency - 3.1 ms eager for one string; the encoder does the same, embedding_service.py).
    # Batches of <= 32 strings and <= 128 tokens are padded to a (batch, width) bucket; padded tokens carry mask 0 and their
    # outputs are never read. Capture failure -> eager, for good.
    _GRAPH_BATCHES = (1, 2, 4, 8, 16, 32)
    _GRAPH_WIDTHS = (16, 32, 64, 128)

    def _logits_to_scores(self, logits):
        # (the pipeline does this softmax in numpy float32 on the host: the same formula, equal to ~1e-7)
        return self.torch.softmax(logits.float(), dim=-1).max(dim=-1)

    def _scores(self, ids, mask, pad):
        """ids, mask: numpy int64 [b, w] -> (probability of the best label, its index) per token, device tensors [b, w]"""
        torch = self.torch
        b, w = ids.shape
        graphs = getattr(self, "_graphs", None)
        if graphs is None and not hasattr(self, "_graphs"):
            graphs = self._graphs = graph_capture.Buckets() if (str(self.device).startswith("cuda") and os.getenv("ICD_NER_GRAPHS", "1") == "1") else None
        if graphs is None or b > self._GRAPH_BATCHES[-1] or w > self._GRAPH_WIDTHS[-1]:
            logits = self.model(input_ids=torch.from_numpy(ids).to(self.device),
                                attention_mask=torch.from_numpy(mask).to(self.device)).logits
            return self._logits_to_scores(logits)
        bb = next(x for x in self._GRAPH_BATCHES if x >= b)
        wb = next(x for x in self._GRAPH_WIDTHS if x >= w)
        entry = graphs.get((bb, wb))
        if entry is None:
            try:
                entry = self._capture(bb, wb)
            except Exception as exc:  # pragma: no cover - depends on the runtime
                gave_up = graphs.failed((bb, wb))   # (this call runs eagerly; the bucket is retried at its next use, a few times)
                logger.warning("HIP graph capture of bucket %s failed (%s): this call runs eagerly%s", (bb, wb), exc,
                               "; the bucket stays eager" if gave_up else "")
                entry = False
            else:
                graphs.put((bb, wb), entry)
        if entry is False:
            logits = self.model(input_ids=torch.from_numpy(ids).to(self.device),
                                attention_mask=torch.from_numpy(mask).to(self.device)).logits
            return self._logits_to_scores(logits)
        g, sids, smask, sscore, slabel = entry
        sids.fill_(pad)
        smask.zero_()
        smask[b:, 0] = 1   # (unused rows: one live token, so no row of the attention is fully masked)
        sids[:b, :w].copy_(torch.from_numpy(ids), non_blocking=True)
        smask[:b, :w].copy_(torch.from_numpy(mask), non_blocking=True)
        g.replay()
        return sscore[:b, :w].clone(), slabel[:b, :w].clone()

    def _capture(self, bb: int, wb: int):
        torch = self.torch
        sids = torch.zeros((bb, wb), dtype=torch.long, device=self.device)
        smask = torch.ones((bb, wb), dtype=torch.long, device=self.device)
        g, (sscore, slabel) = graph_capture.capture(torch, lambda: self._logits_to_scores(self.model(input_ids=sids, attention_mask=smask).logits))
        return g, sids, smask, sscore, slabel

    # -- forward: one padded batch per max_batch strings; softmax, best label and its probability per token on the device
    def _forward(self, encoded):
        """-> per string (label index per token, its float32 probability per token), as Python lists"""
        torch = self.torch
        out = [None] * len(encoded)
        order = sorted(range(len(encoded)), key=lambda i: len(encoded[i][0]))   # length-sorted: little padding per batch
        pad = getattr(self.tokenizer, "pad_token_id", None)
        pad = 0 if pad is None else pad
        if self._small is not None and encoded and self._small.fits([len(e[0]) for e in encoded]):
            with torch.no_grad():
                _, hidden = self._small.encode([e[0] for e in encoded], to_device=True, hidden=True)
                score, label = torch.softmax(self.model.classifier(hidden).float(), dim=-1).max(dim=-1)
                both = torch.stack([score.double(), label.double()]).cpu().numpy()   # (one copy, one synchronisation; exact for both)
                score, label = both[0].astype(np.float32), both[1].astype(np.int64)
            a = 0
            for i, e in enumerate(encoded):
                b = a + len(e[0])
                out[i] = (label[a:b].tolist(), score[a:b])
                a = b
            return out
        if self._packed is not None and len(order) > 32 and self.max_batch > 32:
            with torch.no_grad():
                rev = order[::-1]                                  # longest first
                s = 0
                while s < len(rev):                                # chunks of at most PACK_TOKENS tokens
                    e, tokens = s, 0
                    while e < len(rev) and (e == s or tokens + len(encoded[rev[e]][0]) <= self.PACK_TOKENS):
                        tokens += len(encoded[rev[e]][0])
                        e += 1
                    idx = rev[s:e]
                    hidden, (lengths, starts, _, _) = self._packed.hidden_states([encoded[i][0] for i in idx], self.device)
                    logits = self.model.classifier(hidden)
                    score, label = torch.softmax(logits.float(), dim=-1).max(dim=-1)
                    score, label = score.cpu().numpy(), label.cpu().numpy()
                    for r, i in enumerate(idx):
                        a, b = int(starts[r]), int(starts[r + 1])
                        out[i] = (label[a:b].tolist(), score[a:b])
                    s = e
            return out
        with torch.no_grad():
            for s in range(0, len(order), self.max_batch):
                idx = order[s:s + self.max_batch]
                width = max(len(encoded[i][0]) for i in idx)
                ids = np.full((len(idx), width), pad, dtype=np.int64)
                mask = np.zeros((len(idx), width), dtype=np.int64)
                for r, i in enumerate(idx):
                    n = len(encoded[i][0])
                    ids[r, :n] = encoded[i][0]
                    mask[r, :n] = 1
                score, label = self._scores(ids, mask, pad)
                score, label = score.cpu().numpy(), label.cpu().numpy()
                for r, i in enumerate(idx):
                    n = len(encoded[i][0])
                    out[i] = (label[r, :n].tolist(), score[r, :n])
        return out

    # -- transformers.pipelines.TokenClassificationPipeline.aggregate(SIMPLE) / group_entities -------------------------
    def _groups(self, enc, labels, scores) -> List[Dict[str, Any]]:
        ids, offsets, tokens, special = enc
        keep = [t for t in range(len(ids)) if not special[t]]
        names = [self.id2label[labels[t]] for t in keep]
        groups, i0 = [], 0

        def tag_of(name):
            return ("B", name[2:]) if name.startswith("B-") else (("I", name[2:]) if name.startswith("I-") else ("I", name))

        def close(i0, i1):   # tokens keep[i0:i1] form one group; "O" groups are dropped by the pipeline: skipped here
            group = names[i0].split("-", 1)[-1]
            if group == "O":
                return
            run = keep[i0:i1]
            # group score: np.mean(np.nanmean([token scores])) of float32 scalars = the float32 mean of the slice
            # (special tokens sit at the ends only: a run is a contiguous slice of the float32 score array)
            groups.append({"entity_group": group, "score": (scores[run[0]:run[-1] + 1].mean() if run[-1] - run[0] + 1 == len(run)
                                                            else np.asarray([scores[t] for t in run], dtype=np.float32).mean()),
                           "word": self._join([tokens[t] for t in run]),
                           "start": int(offsets[run[0]][0]), "end": int(offsets[run[-1]][1])})
        for i in range(1, len(keep)):
            bi, tag = tag_of(names[i])
            if not (tag == tag_of(names[i - 1])[1] and bi != "B"):
                close(i0, i)
                i0 = i
        if keep:
            close(i0, len(keep))
        return groups

    def __call__(self, texts: Sequence[str]) -> List[List[Dict[str, Any]]]:
        encoded = self._encode_all(texts)
        return [self._groups(e, lab, sc) for e, (lab, sc) in zip(encoded, self._forward(encoded))]


class MedicalNERService:
    def __init__(self, model_name: str = None, use_model: bool = None):
        if model_name is None:
            model_name = os.getenv("MEDICAL_NER_MODEL", DEFAULT_NER_MODEL)
        if use_model is None:
            use_model = os.getenv("USE_MEDICAL_NER_MODEL", "true").lower() == "true"
        self.model_name = model_name
        self.use_model = use_model
        self.ner_pipeline = None      # a _TokenClassifier when the model is loaded (callable on a list of strings)
        self.model = None
        self.tokenizer = None
        self.synthetic = False
        self.entity_filter = DiagnosisEntityFilter()
        self.entity_type_mapping = dict(ENTITY_TYPE_MAPPING)
        if self.use_model:
            self._init_ner_model()
        else:
            self._init_fallback_patterns()

    # ---- model ---------------------------------------------------------------------------------------------------
    def _init_ner_model(self):
        try:
            import torch
            from transformers import AutoModelForTokenClassification, AutoTokenizer
            device = "cuda" if torch.cuda.is_available() else "cpu"
            try:
                self.tokenizer = AutoTokenizer.from_pretrained(self.model_name, local_files_only=True)
                self.model = AutoModelForTokenClassification.from_pretrained(self.model_name, local_files_only=True)
                id2label = self.model.config.id2label
            except Exception as exc:
                if os.getenv("ICD_NER_ALLOW_SYNTHETIC", "0") != "1":
                    raise
                logger.warning("NER model %s not resolvable offline (%s): using SYNTHETIC weights", self.model_name,
                               type(exc).__name__)
                self.model, self.tokenizer, id2label = _synthetic_classifier()
                self.synthetic = True
            self.ner_pipeline = _TokenClassifier(self.model, self.tokenizer, id2label, device)
        except Exception as exc:
            logger.error("NER model load failed: %s; falling back to the rules", exc)
            self.use_model = False
            self.ner_pipeline = self.model = self.tokenizer = None
            self._init_fallback_patterns()

    def _init_fallback_patterns(self):
        self.medical_patterns = {k: list(v) for k, v in MEDICAL_PATTERNS.items()}
        self.stop_words = set(STOP_WORDS)
        self.meaningless_phrases = set(MEANINGLESS_PHRASES)

    # ---- extraction ------------------------------------------------------------------------------------------------
    def extract_medical_entities(self, text: str, filter_drugs: bool = True) -> Dict[str, List[Dict[str, Any]]]:
        if not text or not text.strip():
            return {}
        return self.extract_medical_entities_batch([text], filter_drugs)[0]

    def extract_medical_entities_batch(self, texts: Sequence[str], filter_drugs: bool = True) -> List[Dict[str, List[Dict[str, Any]]]]:
        """extract_medical_entities for many strings with ONE classifier batch (additive). Empty strings give {}."""
        live = [i for i, t in enumerate(texts) if t and t.strip()]
        results: List[Dict[str, List[Dict[str, Any]]]] = [{} for _ in texts]
        groups = None
        if self.use_model and self.ner_pipeline and live:
            try:
                groups = self.ner_pipeline([texts[i] for i in live])
            except Exception as exc:
                logger.warning("NER model failed, using the rules: %s", exc)
        for j, i in enumerate(live):
            text = texts[i]
            entities = None
            if groups is not None:
                try:
                    entities = self._entities_from_groups(groups[j])
                except Exception as exc:
                    logger.warning("NER model failed, using the rules: %s", exc)
            if entities is None:
                entities = self._extract_entities_with_rules(text)
            if filter_drugs:
                entities = self.entity_filter.filter_entities(entities, text)
            results[i] = entities
        return results

    def _extract_entities_with_model(self, text: str) -> Dict[str, List[Dict[str, Any]]]:
        return self._entities_from_groups(self.ner_pipeline([text])[0])

    def _entities_from_groups(self, model_entities: List[Dict[str, Any]]) -> Dict[str, List[Dict[str, Any]]]:
        entities: Dict[str, List[Dict[str, Any]]] = {}
        for ent in model_entities:
            word = ent["word"].replace(" ", "").replace("##", "")
            label = ent["entity_group"] if "entity_group" in ent else ent["entity"]
            confidence = ent["score"]
            if not self._is_valid_model_entity(word, confidence):
                continue
            entities.setdefault(self.entity_type_mapping.get(label, "other"), []).append({
                "text": word, "start": ent.get("start", 0), "end": ent.get("end", len(word)), "confidence": confidence,
                "original_label": label, "source": "model"})
        for kind in entities:
            entities[kind] = self._deduplicate_entities(entities[kind])
        return entities

    def _extract_entities_with_rules(self, text: str) -> Dict[str, List[Dict[str, Any]]]:
        if not hasattr(self, "medical_patterns"):   # (a model failure at run time: the reference raises here too - AttributeError)
            raise AttributeError("'MedicalNERService' object has no attribute 'medical_patterns'")
        entities = {}
        for kind, patterns in self.medical_patterns.items():
            found = []
            for pattern in patterns:
                for m in re.finditer(pattern, text):
                    word = m.group().strip()
                    if self._is_valid_entity(word):
                        found.append({"text": word, "start": m.start(), "end": m.end(),
                                      "confidence": self._calculate_entity_confidence(word, kind),
                                      "pattern": pattern, "source": "rules"})
            entities[kind] = self._deduplicate_entities(found)
        return entities

    def _is_valid_model_entity(self, entity_text: str, confidence: float) -> bool:
        if not entity_text or len(entity_text) < 2:
            return False
        if confidence < float(os.getenv("MEDICAL_NER_MIN_CONFIDENCE", "0.5")):
            return False
        return not (hasattr(self, "stop_words") and entity_text in self.stop_words)

    def _is_valid_entity(self, entity_text: str) -> bool:
        if not entity_text or len(entity_text) < 2:
            return False
        if entity_text in self.stop_words or entity_text in self.meaningless_phrases:
            return False
        return not re.match(r"^[\d\s\-+.]+$", entity_text)

    def _calculate_entity_confidence(self, entity_text: str, entity_type: str) -> float:
        confidence = 0.5
        if len(entity_text) >= 4:
            confidence += 0.1
        if len(entity_text) >= 6:
            confidence += 0.1
        if any(m in entity_text for m in _TYPE_MARKERS.get(entity_type, ())):
            confidence += 0.2
        if entity_type == "disease" and any(p in entity_text for p in ("急性", "慢性", "原发性")):
            confidence += 0.1
        return min(confidence, 1.0)

    def _deduplicate_entities(self, entities: List[Dict]) -> List[Dict]:
        """Overlapping spans: the first by (start, -confidence) stays unless a later one is more confident (:322-351)."""
        if not entities:
            return []
        entities.sort(key=lambda e: (e["start"], -e["confidence"]))
        kept: List[Dict] = []
        for ent in entities:
            clash = next((o for o in kept if ent["start"] < o["end"] and ent["end"] > o["start"]), None)
            if clash is None:
                kept.append(ent)
            elif ent["confidence"] > clash["confidence"]:
                kept.remove(clash)
                kept.append(ent)
        return sorted(kept, key=lambda e: e["confidence"], reverse=True)

    # ---- summaries -------------------------------------------------------------------------------------------------
    def identify_diagnosis_keywords(self, text: str) -> List[str]:
        entities = self.extract_medical_entities(text)
        disease_min, symptom_min = (0.5, 0.6) if self.use_model else (0.6, 0.7)
        words = [e["text"] for e in entities.get("disease", []) if e["confidence"] > disease_min]
        if not words:
            words = [e["text"] for e in entities.get("symptom", []) if e["confidence"] > symptom_min]
        return words

    def get_model_info(self) -> Dict[str, Any]:
        try:
            import torch
            gpu = torch.cuda.is_available()
            count = torch.cuda.device_count() if gpu else 0
        except ImportError:
            gpu, count = False, 0
        return {"model_name": self.model_name, "use_model": self.use_model, "model_loaded": self.ner_pipeline is not None,
                "entity_types": list(self.entity_type_mapping.keys()) if self.use_model else list(self.medical_patterns.keys()),
                "fallback_available": hasattr(self, "medical_patterns"), "gpu_available": gpu, "gpu_device_count": count,
                "device": "GPU" if gpu and self.use_model else "CPU"}

    def get_entity_summary(self, text: str) -> Dict[str, Any]:
        entities = self.extract_medical_entities(text)
        high = 0.8 if self.use_model else 0.7
        primary = 0.5 if self.use_model else 0.6
        return {
            "total_entities": sum(len(v) for v in entities.values()),
            "entity_types": list(entities.keys()),
            "high_confidence_entities": [
                {"type": kind, "text": e["text"], "confidence": e["confidence"], "source": e.get("source", "unknown")}
                for kind, items in entities.items() for e in items if e["confidence"] > high],
            "primary_diagnosis_candidates": [e["text"] for e in entities.get("disease", [])[:3] if e["confidence"] > primary],
            "extraction_method": "model" if self.use_model and self.ner_pipeline else "rules",
            "model_info": self.get_model_info(),
        }

    def get_filter_stats(self, text: str) -> Dict[str, Any]:
        return self.entity_filter.get_filter_stats(self.extract_medical_entities(text, filter_drugs=False),
                                                   self.extract_medical_entities(text, filter_drugs=True))


def _synthetic_classifier(seed: int = 20251004):
    """A seeded random-init BERT-base token classifier with the reference model's label inventory (B-/I- per type + O)
    and a character tokenizer: the shape of the real model, for throughput measurements only."""
    import torch
    from transformers import BertConfig, BertForTokenClassification
    labels = ["O"] + [f"{p}-{name}" for name in ENTITY_TYPE_MAPPING for p in ("B", "I")]
    cfg = BertConfig(vocab_size=21128, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                     intermediate_size=3072, max_position_embeddings=512, num_labels=len(labels),
                     id2label=dict(enumerate(labels)), label2id={l: i for i, l in enumerate(labels)})
    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = BertForTokenClassification(cfg)
    torch.random.set_rng_state(state)
    return model, _CharOffsetTokenizer(cfg.vocab_size, cfg.max_position_embeddings), cfg.id2label
