import time, torch, os, sys
sys.path.insert(0, os.getcwd())
os.environ["ICD_EMBEDDING_ALLOW_SYNTHETIC"]="1"; os.environ["EMBEDDING_MODEL_NAME"]="shibing624/text2vec-base-chinese"
from rag_project_icd10_amd.services.embedding_service import EmbeddingService
es = EmbeddingService()
m = es.model
dev = es.device
for (B, W) in ((1, 16), (32, 16), (256, 16), (256, 32)):
    tok = torch.randint(1000, 20000, (B, W), device=dev); mask = torch.ones((B, W), dtype=torch.long, device=dev)
    for _ in range(3): out = m(tok, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m(tok, mask)
    torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): m(tok, mask)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            gout = m(tok, mask)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); gr = (time.perf_counter() - t0) / 10
        diff = (gout - out).abs().max().item()
        print(f"B={B} W={W}: eager {eager*1e3:.2f} ms, graph {gr*1e3:.2f} ms, max diff {diff:.2e}")
    except Exception as e:
        print(f"B={B} W={W}: eager {eager*1e3:.2f} ms, graph capture failed: {type(e).__name__}: {str(e)[:200]}")
