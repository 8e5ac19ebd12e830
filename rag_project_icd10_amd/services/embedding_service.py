"""Text -> unit vector. Drop-in for the reference's services/embedding_service.py.

Same public surface (:75-149): encode_single, encode_batch, encode_icd_record, encode_query,
get_model_info, test_embedding; attributes .config .model .device (read by main.py:168,266).
Same text handling, byte for byte (:68-73,:119): `encode_query` prepends "query: " (used for corpus
rows at build time AND for queries), everything else prepends "passage: " unless the text already
starts with "query:" / "passage:".

The arithmetic the reference delegates to sentence-transformers (un-vendored; `>=4.1.0`,
requirements.txt:7) is restated on PyTorch-ROCm: BERT encoder forward -> mean pooling over the
attention mask -> L2 normalisation -> float32 (the published SentenceTransformer.encode contract with
normalize_embeddings=True). Differences by design: texts are really batched, and batches can stay on the
GPU (`encode_query_batch(..., to_device=True)`) so the search kernel reads them without a host round trip.

ONE arithmetic whatever the call shape (round 6): the reference embeds corpus rows and queries through the same one-string
call (tools/build_database.py:217-222 is a loop of encode_query) - identical text, identical vector. Here one string per call
runs the hand-written fp32 forward of csrc/encoder_small.hpp, and any batch the SAME arithmetic in large tiles
(csrc/encoder_big.hpp, icd_encoder_encode_many): row i of a batch is bit for bit encode_query of string i. The packed
split-bf16 forward on PyTorch (_PackedBert: 2.4 x the throughput, vectors ~1.2e-6 off, near-tied hits may swap) is opt-in:
ICD_EMBEDDING_BATCH=fast or fast=True; models the hand-written encoder is not instantiated for keep the framework's forward
(get_model_info()["batch_arithmetic"] says which).

Weights: `EMBEDDING_MODEL_NAME` is resolved locally only (no network in this deployment). If it
cannot be resolved the constructor raises, exactly like the reference does on a failed model load
(:64-66), unless synthetic weights are explicitly allowed (`allow_synthetic=True` or
ICD_EMBEDDING_ALLOW_SYNTHETIC=1): then a seeded random-init BERT-base of text2vec-base-chinese's
shape and a character-level tokenizer are used and `get_model_info()["synthetic"]` is True.
Encoder NUMERICAL parity with the reference is unpinned.
"""
from __future__ import annotations

import itertools
import logging
import os
from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .. import graph_capture
from ..dotenv_lite import load_dotenv

load_dotenv()   # the reference does this at import (python-dotenv); existing environment variables win
logger = logging.getLogger(__name__)

DEFAULT_MODEL = "intfloat/multilingual-e5-large-instruct"  # reference default (:26)
_PREFIXES = ("query:", "passage:")


class _CharTokenizer:
    """Stand-in for the WordPiece vocabulary when no tokenizer files are available: one token per
    character (what bert-base-chinese does for CJK), deterministic ids in [1000, vocab)."""

    cls_id, sep_id, pad_id = 101, 102, 0

    def __init__(self, vocab_size: int, max_len: int):
        self.vocab_size, self.max_len = vocab_size, max_len

    def encode(self, text: str) -> List[int]:
        ids = [1000 + (ord(ch) % (self.vocab_size - 1000)) for ch in text.lower() if not ch.isspace()]
        return [self.cls_id] + ids[: self.max_len - 2] + [self.sep_id]


class _Encoder(torch.nn.Module):
    """HF BertModel + pooling (masked mean, or the [CLS] token) + L2 normalisation."""

    def __init__(self, bert, pooling: str = "mean"):
        super().__init__()
        self.bert = bert
        self.pooling = pooling

    @torch.no_grad()
    def forward(self, input_ids, attention_mask):
        hidden = self.bert(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        if self.pooling == "cls":
            pooled = hidden[:, 0]
        else:
            mask = attention_mask.unsqueeze(-1).to(hidden.dtype)
            pooled = (hidden * mask).sum(1) / mask.sum(1).clamp(min=1e-9)
        return torch.nn.functional.normalize(pooled.float(), p=2, dim=1)


class _PackedBert:
    """The same BERT encoder arithmetic over PACKED tokens: no padding anywhere but inside the attention.

    A padded batch spends its GEMMs on pad tokens (1 000 diagnosis strings: 25 776 padded against 18 290 real tokens at 256
    per sub-batch) and cuts them into sub-batches too small to fill the chip (M = 7 k rows: 100 TFLOP/s fp32 against
    135 at M = 20 k). Here every Linear / LayerNorm / GELU of the encoder runs once over [T, hidden] for ALL tokens of a
    chunk, Q / K / V are one fused GEMM, and only the attention itself sees a padded view: the sequences, sorted by
    length, are cut into a few groups of similar length, each gathered to [n_g, L_g] (pads read a zero row), attended
    with a key mask, and scattered back (profiles/r03_encoder_packed.log). Same weights as the module it is built from
    (shared storage, except the fused QKV copy); post-LN BERT / RoBERTa / XLM-R encoders with absolute positions and erf-GELU
    only - anything else keeps the padded HF forward. Arithmetic restated from transformers' BertModel (BertEmbeddings: (word + type) +
    position -> LayerNorm; BertSelfAttention via scaled_dot_product_attention; BertSelfOutput / BertOutput: dense ->
    LayerNorm(x + residual)), the published architecture the reference reaches through sentence-transformers.
    """

    GROUP_RATIO = 0.75   # a group of sequences for the attention: lengths within this factor of its longest
    NATIVE_MAX_LEN = 512  # tokens per sequence icd_packed_attention takes (csrc/attention_kernel.hpp ATT_MAX_SEQ)
    MAX_GROUPS = 12

    @staticmethod
    def supported(bert) -> bool:
        cfg = bert.config
        return (type(bert).__name__ in ("BertModel", "XLMRobertaModel", "RobertaModel") and getattr(cfg, "position_embedding_type", None) in (None, "absolute")
                and getattr(cfg, "hidden_act", "gelu") == "gelu" and not getattr(cfg, "is_decoder", False))

    def __init__(self, bert):
        self.bert = bert
        cfg = bert.config
        self.heads = int(cfg.num_attention_heads)
        self.hidden = int(cfg.hidden_size)
        self.eps = float(cfg.layer_norm_eps)
        # RoBERTa-family embeddings (the reference's default checkpoint, multilingual-e5, is XLM-R) number positions from
        # padding_idx + 1 (create_position_ids_from_input_ids); BERT from 0
        self.pos_offset = int(bert.embeddings.padding_idx) + 1 if type(bert).__name__ != "BertModel" else 0
        # the attention itself over packed tokens (sequences of up to 512 tokens): one hand-written HIP launch per layer
        # (csrc/attention_kernel.hpp through the C ABI) instead of gather -> SDPA -> scatter per group of similar length.
        # fp32 on a GPU with 64-wide heads only; ICD_EMBEDDING_NATIVE_ATTENTION=0 keeps SDPA everywhere.
        self.native_attention = None
        if os.getenv("ICD_EMBEDDING_NATIVE_ATTENTION", "1") == "1" and self.hidden // self.heads == 64:
            try:
                from .. import _native
                _native.load_library()
                self.native_attention = _native.packed_attention
            except Exception as exc:   # (a CPU-only install: the library is not built)
                logger.debug("native packed attention unavailable: %s", exc)
        self.layers = []
        for l in bert.encoder.layer:
            a = l.attention.self
            self.layers.append({
                "wqkv": torch.cat([a.query.weight, a.key.weight, a.value.weight], 0).detach().contiguous(),
                "bqkv": torch.cat([a.query.bias, a.key.bias, a.value.bias], 0).detach().contiguous(),
                "attn_out": l.attention.output, "inter": l.intermediate.dense, "out": l.output,
            })
        # The four Linear layers of a block as SPLIT-bf16 GEMMs on the bf16 MFMA (ICD_EMBEDDING_GEMM=bf16x3, the default for fp32
        # weights on a GPU; =fp32 keeps rocBLAS fp32 GEMMs): x = x_hi + x_lo and W = W_hi + W_lo in bf16 (16 mantissa bits
        # each way), y = x_hi W_hi + x_hi W_lo + x_lo W_hi + b accumulated in fp32 inside ONE GEMM over the concatenated K
        # ([x_hi | x_hi | x_lo | 1 1 0..] @ [W_hi; W_lo; W_hi; b_hi; b_lo; 0..], fp32 output; the A operand - GELU included - by one
        # native pass, icd_split_bf16x3) - the dropped x_lo W_lo term is 2^-16 relative. Measured
        # against the fp32 forward of the same weights: embeddings within 5.3e-7, cosines within 2.8e-7 (the fp32 summation
        # order alone moves them 1.4e-7), at 3 / 16 of the fp32 MFMA's time per FLOP. Needs torch.mm(out_dtype=...).
        self.split_gemm = None   # decided at the first forward on a device (split_weights)

    def _split_ready(self, device) -> bool:
        if self.split_gemm is not None:
            return self.split_gemm
        ok = (os.getenv("ICD_EMBEDDING_GEMM", "bf16x3") == "bf16x3" and str(device).startswith("cuda")
              and self.bert.embeddings.word_embeddings.weight.dtype == torch.float32)
        if ok:
            try:
                from .. import _native
                _native.load_library()
                self._split_a = _native.split_bf16x3

                def split_w(w, b):   # [N, K] fp32, [N] -> [3K + 64, N] bf16: W_hi^T, W_lo^T, W_hi^T stacked along K, then b_hi, b_lo, 0 x 62
                    def hl(t):
                        hi = t.to(torch.bfloat16)
                        return hi, (t - hi.float()).to(torch.bfloat16)
                    whi, wlo = hl(w.detach())
                    bhi, blo = hl(b.detach())
                    pad = torch.zeros((_native.SPLIT_TAIL - 2, w.shape[0]), dtype=torch.bfloat16, device=w.device)
                    return torch.cat([whi.t(), wlo.t(), whi.t(), bhi[None, :], blo[None, :], pad], 0).contiguous()
                for l in self.layers:
                    l["s_qkv"] = split_w(l["wqkv"], l["bqkv"])
                    l["s_attn_out"] = split_w(l["attn_out"].dense.weight, l["attn_out"].dense.bias)
                    l["s_inter"] = split_w(l["inter"].weight, l["inter"].bias)
                    l["s_out"] = split_w(l["out"].dense.weight, l["out"].dense.bias)
                probe = torch.mm(self._split_a(torch.zeros((2, 64), dtype=torch.float32, device=device)),
                                 torch.zeros((3 * 64 + _native.SPLIT_TAIL, 16), dtype=torch.bfloat16, device=device), out_dtype=torch.float32)
                ok = probe.dtype == torch.float32
                self._mm_out = False
                try:   # (the result straight into a slice of the caller's buffer, where this torch takes out= next to out_dtype=)
                    buf = torch.empty((3, 16), dtype=torch.float32, device=device)
                    torch.mm(self._split_a(torch.zeros((2, 64), dtype=torch.float32, device=device)),
                             torch.zeros((3 * 64 + _native.SPLIT_TAIL, 16), dtype=torch.bfloat16, device=device), out_dtype=torch.float32, out=buf[:2])
                    self._mm_out = True
                except Exception:
                    pass
            except Exception as exc:   # (a torch without mm(out_dtype=...), or no native library: the fp32 GEMMs stay)
                logger.info("split-bf16 GEMMs unavailable (%s): fp32 GEMMs", exc)
                ok = False
        self.split_gemm = bool(ok)
        return self.split_gemm

    def _lin3(self, x, w3, gelu=False, out=None):
        """y = act(x) W^T + b through ONE bf16 GEMM with fp32 accumulation and output: [x_hi | x_hi | x_lo | 1 1 0...] @ [W_hi; W_lo; W_hi; b_hi; b_lo; 0...]
        (the A operand by one native pass, csrc/attention_kernel.hpp split_bf16x3_kernel)"""
        if out is not None and self._mm_out:
            return torch.mm(self._split_a(x, gelu), w3, out_dtype=torch.float32, out=out)
        y = torch.mm(self._split_a(x, gelu), w3, out_dtype=torch.float32)
        if out is not None:
            out.copy_(y)
            return out
        return y

    @classmethod
    def plan_groups(cls, lengths):
        """lengths sorted descending -> [(first sequence, count, longest)]: greedy cuts where a sequence falls below
        GROUP_RATIO of the group's longest; the ratio loosens until at most MAX_GROUPS remain"""
        ratio = cls.GROUP_RATIO
        while True:
            groups, start = [], 0
            for i in range(1, len(lengths) + 1):
                if i == len(lengths) or lengths[i] < ratio * lengths[start]:
                    groups.append((start, i - start, lengths[start]))
                    start = i
            if len(groups) <= cls.MAX_GROUPS or ratio <= 0.05:
                return groups
            ratio *= 0.8

    @torch.no_grad()
    def hidden_states(self, ids_sorted, device):
        """ids_sorted: token id lists, longest first. Returns (last hidden state of every token, packed [T, hidden], in
        that order; the plan: lengths, first packed row of every sequence, attention groups)."""
        F = torch.nn.functional
        lengths = [len(x) for x in ids_sorted]
        n, T = len(lengths), sum(lengths)
        lens_np = np.asarray(lengths, dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(lens_np)])
        flat = np.fromiter(itertools.chain.from_iterable(ids_sorted), dtype=np.int64, count=T)
        pos = np.arange(T, dtype=np.int64) - np.repeat(starts[:-1], lens_np) + self.pos_offset
        dtype = self.bert.embeddings.word_embeddings.weight.dtype

        # EVERY host -> device copy of the chunk happens here, before its first kernel: a copy from pageable memory holds
        # the host until the stream has reached it - behind the encoder's kernels that is the whole forward (the caller
        # then cannot prepare the search while the GPU encodes: 51 -> 44 ms per 1 000 strings end to end)
        def up(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=True)
        # sequences of <= NATIVE_MAX_LEN tokens (sorted: the tail) go to the native kernel; only longer ones form SDPA groups
        use_native = (self.native_attention is not None and str(device).startswith("cuda") and dtype == torch.float32)
        n_long = int(np.searchsorted(-lens_np, -self.NATIVE_MAX_LEN, side="left")) if use_native else n   # sequences beyond the kernel's reach
        native = None
        if use_native and n_long < n:
            native = (up((starts[n_long:]).astype(np.int32)), n - n_long, int(lens_np[n_long]))
        def make_groups(upto, with_bias):
            out = []
            for first, count, longest in (self.plan_groups(lengths[:upto]) if upto else []):
                col = np.arange(longest, dtype=np.int64)[None, :]
                key = col < lens_np[first:first + count, None]
                grid = np.where(key, starts[first:first + count, None] + col, T)   # pads point at the zero row behind the tokens
                # (the key mask as the additive bias SDPA would make of a boolean one - once per group, not once per layer;
                #  a group without pads needs none)
                bias = None if (not with_bias or bool(key.all())) else up(np.where(key, 0.0, -np.inf).astype(np.float32)).to(dtype)[:, None, None, :]
                lens_t = up(np.maximum(lens_np[first:first + count], 1e-9).astype(np.float32)).to(dtype)
                out.append((count, longest, up(grid.reshape(-1)), bias, lens_t))
            return out
        groups = make_groups(n_long, True)                                  # attention through SDPA: these sequences
        pool_groups = groups if n_long == n else make_groups(n, False)      # mean pooling: every sequence
        ids_t, pos_t, first_rows = up(flat), up(pos), up(starts[:-1])
        emb = self.bert.embeddings
        x = emb.word_embeddings(ids_t) + emb.token_type_embeddings.weight[0]
        x = x + emb.position_embeddings(pos_t)
        x = emb.LayerNorm(x)
        H, nh = self.hidden, self.heads
        dh = H // nh
        qkv = torch.zeros((T + 1, 3 * H), dtype=x.dtype, device=device)   # row T: the pads' zero row
        ctx = torch.empty((T + 1, H), dtype=x.dtype, device=device)       # row T: where the pads' outputs land
        split = self._split_ready(device)
        for l in self.layers:
            if split:
                self._lin3(x, l["s_qkv"], out=qkv[:T])
            else:
                torch.addmm(l["bqkv"], x, l["wqkv"].t(), out=qkv[:T])
            for count, longest, grid, bias, _ in groups:
                g = qkv.index_select(0, grid).view(count, longest, 3, nh, dh)
                q, k, v = (g[:, :, i].transpose(1, 2) for i in range(3))
                o = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
                ctx.index_copy_(0, grid, o.transpose(1, 2).reshape(count * longest, H))
            if native is not None:
                self.native_attention(qkv, native[0], native[1], nh, native[2], ctx)
            if split:
                x = l["attn_out"].LayerNorm(self._lin3(ctx[:T], l["s_attn_out"]).add_(x))
                x = l["out"].LayerNorm(self._lin3(self._lin3(x, l["s_inter"]), l["s_out"], gelu=True).add_(x))
            else:
                x = l["attn_out"].LayerNorm(l["attn_out"].dense(ctx[:T]) + x)
                x = l["out"].LayerNorm(l["out"].dense(F.gelu(l["inter"](x))) + x)
        return x, (lengths, starts, pool_groups, first_rows)

    @torch.no_grad()
    def forward(self, ids_sorted, device, pooling: str):
        """ids_sorted: token id lists, longest first. Returns the pooled, UN-normalised [n, hidden] in that order."""
        x, (lengths, starts, groups, first_rows) = self.hidden_states(ids_sorted, device)
        n, H = len(lengths), self.hidden
        if pooling == "cls":
            return x.index_select(0, first_rows)
        xe = torch.cat([x, x.new_zeros((1, H))], 0)
        pooled = torch.empty((n, H), dtype=x.dtype, device=device)
        first = 0
        for count, longest, grid, _, lens_t in groups:
            pooled[first:first + count] = xe.index_select(0, grid).view(count, longest, H).sum(1) / lens_t[:, None]
            first += count
        return pooled


class UnsupportedPoolingError(ValueError):
    """the checkpoint asks for a sentence-transformers pooling mode this restatement does not implement"""


def _sentence_transformers_config(name: str) -> Dict[str, Any]:
    """What SentenceTransformer(name) would read next to the weights: `sentence_bert_config.json` (max_seq_length) and the
    pooling module's config.json (modules.json names its directory, `1_Pooling` by convention). Returns {} when the
    model is not a local sentence-transformers checkpoint. Raises on a pooling mode this restatement does not implement
    (the embeddings would silently differ from the reference's)."""
    import json
    base = name if os.path.isdir(name) else None
    if base is None:
        try:
            from huggingface_hub import snapshot_download
            base = snapshot_download(name, local_files_only=True)
        except Exception:
            return {}
    out: Dict[str, Any] = {}
    sb = os.path.join(base, "sentence_bert_config.json")
    if os.path.exists(sb):
        with open(sb, encoding="utf-8") as f:
            msl = json.load(f).get("max_seq_length")
        if msl:
            out["max_seq_length"] = int(msl)
    pool_dir = "1_Pooling"
    mods = os.path.join(base, "modules.json")
    if os.path.exists(mods):
        with open(mods, encoding="utf-8") as f:
            for m in json.load(f):
                if str(m.get("type", "")).endswith("Pooling"):
                    pool_dir = m.get("path", pool_dir)
    pc = os.path.join(base, pool_dir, "config.json")
    if os.path.exists(pc):
        with open(pc, encoding="utf-8") as f:
            cfg = json.load(f)
        on = sorted(k for k, v in cfg.items() if k.startswith("pooling_mode_") and v is True)
        if on == ["pooling_mode_mean_tokens"]:
            out["pooling"] = "mean"
        elif on == ["pooling_mode_cls_token"]:
            out["pooling"] = "cls"
        else:
            raise UnsupportedPoolingError(f"unsupported sentence-transformers pooling {on} in {pc}: only mean_tokens and cls_token are implemented")
    return out


class EmbeddingService:
    def __init__(self, allow_synthetic: Optional[bool] = None, device: Optional[str] = None):
        self.config = self._load_config()
        if device is not None:
            self.config["embedding"]["device"] = device
        if allow_synthetic is None:
            allow_synthetic = os.getenv("ICD_EMBEDDING_ALLOW_SYNTHETIC", "0") == "1"
        self._allow_synthetic = allow_synthetic
        self.model = None
        self.synthetic = False
        self.device = self._get_device()
        self._load_model()

    # ---- configuration (reference :22-45) --------------------------------------------------------------
    def _load_config(self) -> Dict[str, Any]:
        return {"embedding": {
            "model_name": os.getenv("EMBEDDING_MODEL_NAME", DEFAULT_MODEL),
            "max_length": 512,
            "batch_size": 32,
            "device": os.getenv("EMBEDDING_DEVICE", "auto"),
        }}

    def _get_device(self) -> str:
        dev = self.config.get("embedding", {}).get("device", "auto")
        if dev != "auto":
            return dev
        if torch.cuda.is_available():
            return "cuda"
        if hasattr(torch.backends, "mps") and torch.backends.mps.is_available():
            return "mps"
        return "cpu"

    def _load_model(self):
        name = self.config.get("embedding", {}).get("model_name", DEFAULT_MODEL)
        if not name:
            raise ValueError("模型名称不能为空")
        from transformers import BertConfig, BertModel
        tok, bert = None, None
        try:
            from transformers import AutoModel, AutoTokenizer
            tok = AutoTokenizer.from_pretrained(name, local_files_only=True)
            bert = AutoModel.from_pretrained(name, local_files_only=True)
            self.max_seq_length = int(min(getattr(tok, "model_max_length", 512), 512))
            st_cfg = _sentence_transformers_config(name)   # raises on an unsupported pooling mode (never synthetic then)
            self.max_seq_length = int(st_cfg.get("max_seq_length", self.max_seq_length))
            pooling = st_cfg.get("pooling", "mean")
        except UnsupportedPoolingError:
            raise
        except Exception as exc:
            if not self._allow_synthetic:
                logger.error("模型加载失败: %s", exc)
                raise
            logger.warning("model %s not resolvable offline (%s): using SYNTHETIC weights", name, type(exc).__name__)
            # shape of shibing624/text2vec-base-chinese (BERT-base, vocab 21128, max_seq_length 128)
            cfg = BertConfig(vocab_size=21128, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                             intermediate_size=3072, max_position_embeddings=512)
            gen_state = torch.random.get_rng_state()
            torch.manual_seed(0)
            bert = BertModel(cfg, add_pooling_layer=False)
            torch.random.set_rng_state(gen_state)
            tok = None
            self.synthetic = True
            self.max_seq_length = 128
            pooling = "mean"
        self._tokenizer = tok
        self._char_tok = _CharTokenizer(bert.config.vocab_size, self.max_seq_length) if tok is None else None
        dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[os.getenv("ICD_EMBEDDING_DTYPE", "fp32")]
        self.pooling = pooling
        self.model = _Encoder(bert, pooling).eval().to(self.device)
        if dtype != torch.float32:
            self.model.bert.to(dtype)
        self._dim = int(bert.config.hidden_size)
        self._graphs = graph_capture.Buckets() if os.getenv("ICD_EMBEDDING_GRAPHS", "1") == "1" else None
        # large batches: the encoder over packed tokens (ICD_EMBEDDING_PACKED=0: the padded HF forward everywhere)
        self._packed = (_PackedBert(self.model.bert) if os.getenv("ICD_EMBEDDING_PACKED", "1") == "1" and _PackedBert.supported(self.model.bert)
                        else None)
        # ONE string per call (the reference's encode_query / encode_single, :97-120) or a request's handful: the hand-written
        # small-input forward, one graph launch (csrc/encoder_small.hpp through the C ABI; ICD_EMBEDDING_SMALL=0: the replayed
        # graph of the framework's forward). fp32 BERT-base shapes on a GPU only; anything else keeps the paths below.
        self._small = None
        # batches beyond one small-input call: "canonical" (default) = the same kernels, cut into calls (bit-identical to one string
        # per call); "fast" = the packed split-bf16 forward
        self._batch_fast = os.getenv("ICD_EMBEDDING_BATCH", "canonical").strip().lower() == "fast"
        if os.getenv("ICD_EMBEDDING_SMALL", "1") == "1" and str(self.device).startswith("cuda") and dtype == torch.float32:
            try:
                from .. import _native
                if _native.SmallEncoder.supported(self.model.bert):
                    self._small = _native.SmallEncoder(self.model.bert)
            except Exception as exc:   # (a CPU-only install, an unsupported architecture: the framework's forward stays)
                logger.info("small-input encoder unavailable (%s): the framework's forward", exc)

    # ---- text preparation (reference :68-73) ------------------------------------------------------------
    def _prepare_text_for_embedding(self, text: str) -> str:
        if not text.startswith(_PREFIXES):
            text = f"passage: {text}"
        return text

    # ---- the forward pass --------------------------------------------------------------------------------
    def _tokenize(self, texts: List[str]) -> List[List[int]]:
        if self._tokenizer is not None:
            enc = self._tokenizer(texts, add_special_tokens=True, truncation=True, max_length=self.max_seq_length)
            return enc["input_ids"]
        return [self._char_tok.encode(t) for t in texts]

    def _encode_prepared(self, texts: List[str], batch_size: int, to_device: bool = False, fast: Optional[bool] = None):
        """texts already carry their prefix. Returns float32 [n, dim] (numpy, or a tensor on self.device)."""
        n = len(texts)
        if n == 0:
            out = torch.empty((0, self._dim), dtype=torch.float32, device=self.device)
            return out if to_device else out.cpu().numpy()
        ids = self._tokenize(texts)
        if self._small is not None:
            lens = [len(x) for x in ids]
            if self._small.fits(lens):
                return self._small.encode(ids, pooling=self.pooling, normalize=True, to_device=to_device)
            # ONE embedding arithmetic whatever the call shape (the canonical one: csrc/encoder_small.hpp). The reference embeds
            # corpus rows and queries through the same one-string call (tools/build_database.py:217-222, :117-120): identical text,
            # identical vector. A batch goes through the same kernels cut into calls by the library - row i is bit for bit what
            # encode_query(texts[i]) returns. `fast` (per call, or ICD_EMBEDDING_BATCH=fast) trades that for the packed split-bf16
            # forward below: ~5 x the throughput, vectors within ~1.2e-6 of the canonical ones.
            if not (self._batch_fast if fast is None else fast) and self._small.fits_each(lens):
                return self._small.encode_many(ids, pooling=self.pooling, normalize=True, to_device=to_device)
        out = torch.empty((n, self._dim), dtype=torch.float32, device=self.device)
        order = sorted(range(n), key=lambda i: -len(ids[i]))  # length buckets: least padding per batch
        if self._packed is not None and batch_size > self._GRAPH_BATCHES[-1] and n > self._GRAPH_BATCHES[-1]:
            # chunks of at most PACK_TOKENS tokens (activation memory: ~40 KB per token in fp32), longest strings first
            s = 0
            while s < n:
                e, tokens = s, 0
                while e < n and (e == s or tokens + len(ids[order[e]]) <= self.PACK_TOKENS):
                    tokens += len(ids[order[e]])
                    e += 1
                idx = order[s:e]
                idx_t = torch.tensor(idx, dtype=torch.long).to(self.device, non_blocking=True)   # (before the forward: see _PackedBert)
                pooled = self._packed.forward([ids[i] for i in idx], self.device, self.pooling)
                out.index_copy_(0, idx_t, torch.nn.functional.normalize(pooled.float(), p=2, dim=1))
                s = e
            return out if to_device else out.cpu().numpy()
        pad = self._tokenizer.pad_token_id if self._tokenizer is not None else _CharTokenizer.pad_id
        for s in range(0, n, batch_size):
            idx = order[s:s + batch_size]
            width = len(ids[idx[0]])
            tok = torch.full((len(idx), width), pad, dtype=torch.long)
            mask = torch.zeros((len(idx), width), dtype=torch.long)
            for r, i in enumerate(idx):
                tok[r, : len(ids[i])] = torch.tensor(ids[i], dtype=torch.long)
                mask[r, : len(ids[i])] = 1
            emb = self._forward(tok, mask)
            out[torch.tensor(idx, device=self.device)] = emb
        return out if to_device else out.cpu().numpy()

    # ---- small batches replay a captured HIP graph ---------------------------------------------------------------
    # The reference encodes ONE string per call (encode_query, :117-120): ~150 tiny kernels whose launch overhead, not
    # their arithmetic, sets the latency (3.1 ms eager vs 1.1 ms replayed on MI355X, profiles/r01_encoder_graph_probe.log).
    # Batches of <= 32 strings are padded to a (batch, width) bucket and replayed; padded tokens carry mask 0 (no
    # effect on attention or pooling). Larger batches are compute-bound and run eagerly. Capture failure -> eager.
    _GRAPH_BATCHES = (1, 2, 4, 8, 16, 32)
    PACK_TOKENS = 65536
    _GRAPH_WIDTHS = (16, 32, 64, 128)

    def _forward(self, tok, mask):
        b, w = tok.shape
        use_graph = (self._graphs is not None and str(self.device).startswith("cuda") and b <= self._GRAPH_BATCHES[-1]
                     and w <= self._GRAPH_WIDTHS[-1])
        if not use_graph:
            return self.model(tok.to(self.device, non_blocking=True), mask.to(self.device, non_blocking=True))
        bb = next(x for x in self._GRAPH_BATCHES if x >= b)
        wb = next(x for x in self._GRAPH_WIDTHS if x >= w)
        entry = self._graphs.get((bb, wb))
        if entry is None:
            try:
                entry = self._capture(bb, wb)
            except Exception as exc:  # pragma: no cover - depends on the runtime
                gave_up = self._graphs.failed((bb, wb))   # (this call runs eagerly; the bucket is retried at its next use, a few times)
                logger.warning("HIP graph capture of bucket %s failed (%s): this call runs eagerly%s", (bb, wb), exc,
                               "; the bucket stays eager" if gave_up else "")
                entry = False
            else:
                self._graphs.put((bb, wb), entry)
        if entry is False:
            return self.model(tok.to(self.device, non_blocking=True), mask.to(self.device, non_blocking=True))
        g, stok, smask, sout = entry
        pad = self._tokenizer.pad_token_id if self._tokenizer is not None else _CharTokenizer.pad_id
        stok.fill_(pad)
        smask.zero_()
        smask[b:, 0] = 1   # (unused rows: one live token, so the pooling never divides by zero)
        stok[:b, :w].copy_(tok, non_blocking=True)
        smask[:b, :w].copy_(mask, non_blocking=True)
        g.replay()
        return sout[:b].clone()

    def _capture(self, bb: int, wb: int):
        dev = self.device
        stok = torch.zeros((bb, wb), dtype=torch.long, device=dev)
        smask = torch.ones((bb, wb), dtype=torch.long, device=dev)
        # (thread-local capture mode under the process-wide capture lock: a worker thread's allocations or copies - the NER job of
        #  the same request - cannot invalidate it, graph_capture.py)
        g, sout = graph_capture.capture(torch, lambda: self.model(stok, smask))
        return g, stok, smask, sout

    # ---- reference API -------------------------------------------------------------------------------------
    def encode_single(self, text: str) -> np.ndarray:
        if not self.model:
            raise RuntimeError("嵌入模型未加载")
        return self._encode_prepared([self._prepare_text_for_embedding(text)], 1)[0]

    def encode_batch(self, texts: List[str], show_progress: bool = True) -> List[List[float]]:
        if not self.model:
            raise RuntimeError("嵌入模型未加载")
        if not texts:
            return []
        prepared = [self._prepare_text_for_embedding(t) for t in texts]
        bs = self.config.get("embedding", {}).get("batch_size", 32)
        # (the reference hands batch_size = 32 to sentence-transformers, which sorts by length and batches: the embedding of
        #  a text does not depend on the batching. More than 32 texts take the packed forward in one piece: 1 000 texts
        #  34 ms instead of 32 replayed sub-batches)
        if self._packed is not None and len(prepared) > self._GRAPH_BATCHES[-1]:
            bs = max(bs, 256)
        return self._encode_prepared(prepared, bs).tolist()

    def encode_icd_record(self, icd_record: Dict[str, Any]) -> np.ndarray:
        name = icd_record.get("preferred_zh", "")
        if not name.strip():
            name = f"ICD代码 {icd_record.get('code', 'unknown')}"
        return self.encode_single(name)

    def encode_query(self, query: str) -> np.ndarray:
        if self.model is None:  # the reference has no guard here (:117-120): same AttributeError
            raise AttributeError("'NoneType' object has no attribute 'encode'")
        return self._encode_prepared([f"query: {query}"], 1)[0]

    def get_model_info(self) -> Dict[str, Any]:
        if not self.model:
            return {"loaded": False}
        return {
            "loaded": True,
            "model_name": self.config.get("embedding", {}).get("model_name"),
            "device": self.device,
            "max_seq_length": self.max_seq_length,
            "embedding_dimension": self._dim,
            "synthetic": self.synthetic,
            "pooling": self.pooling,
            "batch_arithmetic": self.batch_arithmetic(),
        }

    def test_embedding(self, test_text: str = "测试文本") -> Dict[str, Any]:
        try:
            emb = self.encode_single(test_text)
            return {"success": True, "embedding_shape": emb.shape, "embedding_type": str(type(emb)),
                    "sample_values": emb[:5].tolist()}
        except Exception as exc:
            return {"success": False, "error": str(exc)}

    # ---- additive batch entry points ---------------------------------------------------
    def encode_query_batch(self, queries: List[str], batch_size: int = 256, to_device: bool = False, fast: Optional[bool] = None):
        """encode_query for many strings at once: float32 [n, dim]; with to_device=True the result
        stays on the GPU for the search kernel. Row i equals encode_query(queries[i]) bit for bit on the canonical path (the
        default where the small-input encoder serves the model); fast=True (or ICD_EMBEDDING_BATCH=fast) takes the packed
        split-bf16 forward instead: vectors within ~1.2e-6, near-tied hits may swap."""
        return self._encode_prepared([f"query: {q}" for q in queries], batch_size, to_device, fast)

    def encode_passage_batch(self, texts: List[str], batch_size: int = 256, to_device: bool = False, fast: Optional[bool] = None):
        return self._encode_prepared([self._prepare_text_for_embedding(t) for t in texts], batch_size, to_device, fast)

    def batch_arithmetic(self) -> str:
        """what a batch of more strings than one small-input call is embedded with: "canonical" (bit-identical to one string per
        call), "fast" (packed split-bf16) or "framework" (no small-input encoder for this model / device)"""
        if self._small is None:
            return "framework"
        return "fast" if self._batch_fast else "canonical"
