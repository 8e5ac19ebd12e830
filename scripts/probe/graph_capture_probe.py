import sys, numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'oracle')
import oracle as orc
from rag_project_icd10_amd._native import IcdIndex, MODE_AUTO
rng = np.random.default_rng(1)
n, dim = 20000, 768
c = rng.standard_normal((n, dim), dtype=np.float32); c /= np.linalg.norm(c, axis=1, keepdims=True)
q = rng.standard_normal((2000, dim), dtype=np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
lv = np.ones(n, np.int32)
idx = IcdIndex(c, lv, max_nq=2000, max_k=10)
dq = torch.from_numpy(q).cuda()
for _ in range(3): idx.search_reweighted(dq, 10)
torch.cuda.synchronize()
try:
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): idx.search_reweighted(dq, 10)
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        out = idx.search_reweighted(dq, 10)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    os_, oi = orc.flat_ip_topk(c, q, 10)
    print('AUTO batch captured and replayed: ids exact', bool(np.array_equal(out[2].cpu().numpy(), oi)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print('replay ms per search', e0.elapsed_time(e1)/20)
except Exception as e:
    print('capture failed:', type(e).__name__, str(e)[:300])
