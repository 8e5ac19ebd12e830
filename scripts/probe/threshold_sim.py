"""Model of the coarse pass's candidate lists: how many scores pass the select per tile and
wave under different threshold-sharing schemes, before any of them is built.

The flat partition of 79 query tiles x 290 corpus tiles over 256 work-groups (U = 90) is reproduced with its lists, their
lengths and the time offsets at which different work-groups sweep the lists of one query tile; every list bootstraps its
threshold over its first tiles (6th best so far), compacts when a lane pair has appended more than ~24 entries and every 24
tiles, and adopts shared thresholds at tile ends. Scores are iid N(0, 1) (2 eps = 0.0024 / 0.0361 sigma).
  cur    the shipped scheme: lists publish the threshold they stand on, everyone adopts the largest
  G r j  lists publish their own r-th best at compactions; everyone adopts (j-th largest of those) - 2 eps (r j >= k)
  ideal  an oracle: the exact running global `rank`-th best over every row any list has seen, `lag` tiles ago, minus 2 eps
  V       the query-global threshold: every list publishes its OWN running k-th best (k = 10, not KP = 16)
         and everyone adopts (the largest published) - 2.02 eps; `every` = 'compaction' (the k-th best is known for free when a
         buffer is compacted) or 'tile' (recomputed at every tile end: a selection over the buffer the kernel would have to add);
         a list still ends on its own KP-th best as its bound, so the certificate tau < s_k - 2 eps is unchanged
Output: (appends per tile and wave, compactions per tile and wave, share of queries whose certificate would fail).
`cur` gives 14.0 appends per tile and wave; the kernel's measured figure is 14.5 registers with a passing lane."""
import numpy as np, sys
rng = np.random.default_rng(1)
CT, U, KP, K = 290, 90, 16, 10
EPS2 = 0.0024/0.0361   # 2 eps in sigma units
NQT = 60               # query tiles simulated (each 'query' = one query of the tile; simulate 8 queries per tile)
QPT = 8

def lists_of_tile(m):
    # returns [(wg_start_time, t0, t1)] for query tile m: runs of work-groups [wU,(w+1)U) cut at list_tiles
    lt = (CT + 1)//2
    out = []
    u0, u1 = m*CT, (m+1)*CT
    w = u0 // U
    while w*U < u1:
        b, e = max(w*U, u0), min((w+1)*U, u1)
        # run inside tile: [b-u0, e-u0); processed at WG time offset b - w*U
        r0 = b - u0
        tstart = b - w*U
        # cut into lists of at most lt tiles
        j = r0
        while j < e - u0:
            j1 = min(e - u0, j + lt)
            out.append((tstart + (j - r0), j, j1))
            j = j1
        w += 1
    return out

def simulate(scheme, r=None, jth=None, epoch=24, seed=0, lag=1, rank=10, every='compaction'):
    rng = np.random.default_rng(seed)
    tot_app = 0; tot_comp = 0; tot_lt = 0; fail = 0; nq = 0
    for m in range(NQT):
        L = lists_of_tile(m)
        P = len(L)
        for q in range(QPT):
            nq += 1
            scores = rng.standard_normal((CT, 128)).astype(np.float32)
            # state per list
            thr = np.full(P, -np.inf); buf = [np.empty(0, np.float32) for _ in range(P)]
            since = np.zeros(P, int); pub_own = np.full(P, -np.inf); pub_r = np.full(P, -np.inf)
            boot_seen = [np.empty(0, np.float32) for _ in range(P)]
            bound = np.full(P, -np.inf)
            T = max(ts + (t1 - t0) for ts, t0, t1 in L)
            seen = np.empty(0, np.float32); hist = []
            for t in range(T):
                new = []
                for li, (ts, t0, t1) in enumerate(L):
                    i = t - ts
                    if i < 0 or i >= t1 - t0: continue
                    ntl = t1 - t0
                    tile = scores[t0 + i]
                    # adopt shared threshold at the tile start (from what was published so far)
                    if scheme == 'cur':
                        s = pub_own.max()
                    elif scheme == 'ideal':
                        tt = t - lag
                        s = hist[tt][rank-1] - EPS2 if tt >= 0 and len(hist[tt]) >= rank else -np.inf
                    elif scheme == 'V':
                        g = pub_r.max()
                        s = max(pub_own.max(), g - EPS2*1.01 if np.isfinite(g) else -np.inf)   # (2.02 eps; the shipped rule stays underneath)
                    else:
                        # G scheme: j-th largest of published own r-th best, minus 2 eps; also own KP-th via pub_own
                        v = np.sort(pub_r)[::-1]
                        g = v[jth-1] - EPS2*1.02 if len(v) >= jth and np.isfinite(v[jth-1]) else -np.inf
                        s = max(g, -np.inf)
                    if s > thr[li]: thr[li] = s
                    boot_tiles = min(8, ntl//3) if ntl >= 6 else 0
                    if i < boot_tiles:
                        boot_seen[li] = np.sort(np.concatenate([boot_seen[li], tile]))[::-1][:6]
                        if len(boot_seen[li]) >= 6 and boot_seen[li][5] > thr[li]: thr[li] = boot_seen[li][5]
                    new.append(tile)
                    passing = tile[tile > thr[li]]
                    tot_app += len(passing); tot_lt += 1
                    buf[li] = np.concatenate([buf[li], passing]); since[li] += len(passing)
                    do = since[li] > 24 or ((i+1) % epoch == 0 and i+1 < ntl and since[li] > 3)
                    if do:
                        tot_comp += 1
                        b = np.sort(buf[li])[::-1]
                        if len(b) > KP:
                            thr[li] = max(thr[li], b[KP-1]); b = b[:KP]   # keep KP, threshold = KP-th (approx of 'cut')
                        buf[li] = b; since[li] = 0
                        if len(b) >= KP: pub_own[li] = max(pub_own[li], b[KP-1])
                        if r is not None and len(b) >= r: pub_r[li] = max(pub_r[li], b[r-1])
                    if scheme in ('cur', 'V'):
                        pub_own[li] = max(pub_own[li], thr[li]) if np.isfinite(thr[li]) else pub_own[li]
                    if scheme == 'V' and every == 'tile' and len(buf[li]) >= K:
                        pub_r[li] = max(pub_r[li], np.sort(buf[li])[::-1][K-1])
                    if i == ntl - 1:
                        b = np.sort(buf[li])[::-1]
                        if len(b) > KP: thr[li] = max(thr[li], b[KP-1]); b = b[:KP]
                        buf[li] = b; bound[li] = thr[li]
                if new: seen = np.sort(np.concatenate([seen] + new))[::-1][:32]
                hist.append(seen.copy())
            allc = np.sort(np.concatenate(buf))[::-1]
            sk = allc[K-1]
            if not (bound.max() < sk - EPS2): fail += 1
    return tot_app/ tot_lt * 32, tot_comp / tot_lt * 32, fail / nq   # per wave-tile (32 queries)


WAVE_TILES = 79 * CT * 4   # wave-tiles of one launch at 10 000 x 37 000 (79 query tiles x 290 corpus tiles x 4 waves)

if __name__ == '__main__':
    print('# (appends per tile and wave, compactions per tile and wave, share of queries failing the certificate)')
    cur = simulate('cur')
    print('cur      ', cur, 'appends per launch %.2f M' % (cur[0] * WAVE_TILES / 1e6))
    if len(sys.argv) > 1 and sys.argv[1] == 'verdict':
        # the query-global (own k-th best - 2.02 eps) threshold, priced before anything is built
        for every in ('compaction', 'tile'):
            v = simulate('V', r=K, every=every)
            print('V own k-th best - 2.02 eps, published at every %-10s' % every, v, 'appends per launch %.2f M (x %.2f of cur)' % (v[0] * WAVE_TILES / 1e6, v[0] / cur[0]))
        o = simulate('ideal', lag=1, rank=16)
        print('oracle: exact global 16th best one tile ago - 2 eps', o, 'appends per launch %.2f M (x %.2f of cur)' % (o[0] * WAVE_TILES / 1e6, o[0] / cur[0]))
        sys.exit(0)
    for r, j in ((10, 1), (5, 2), (4, 3), (3, 4), (2, 5)):
        print('G r=%d j=%d' % (r, j), simulate('G', r, j))
    for lag in (1, 4, 12, 24):   # (rank 10 with no safety margin: the failing share is the model's, a real scheme would use rank 16)
        print('ideal global 10th lag', lag, simulate('ideal', lag=lag))
    print('ideal global 16th lag 4', simulate('ideal', lag=4, rank=16))
    print('ideal global 24th lag 4', simulate('ideal', lag=4, rank=24))
