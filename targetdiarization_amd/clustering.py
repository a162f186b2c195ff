"""HDBSCAN* on the host for the few-hundred speaker embeddings of the path.

The reference clusters per-clip target embeddings with `hdbscan.HDBSCAN(min_cluster_size=2, metric="euclidean")` and
drops the noise label (TargetASR.py:236-243); BASELINE's config 5 all-gathers the `[n*2,192]` embedding block "for
clustering".  The `hdbscan` package is third-party and absent, so the published algorithm is restated here (Campello,
Moulavi, Sander 2013; McInnes, Healy, Astels 2017 — the package's `generic`/`prims` path): core distances -> mutual
reachability -> minimum spanning tree -> single-linkage hierarchy -> condensed tree (min_cluster_size) -> excess-of-mass
selection (no single-cluster result) -> labels, noise = -1.  Dense O(n^2) arithmetic in float64: n is the number of
clips of one speaker or the utterances of a batch, not audio frames.  Parity unpinned against the `hdbscan` package
itself; tests compare with `sklearn.cluster.HDBSCAN` (the same algorithm) where scikit-learn is importable.
"""
from __future__ import annotations

import numpy as np


def _mst_prim(W: np.ndarray) -> np.ndarray:
    """minimum spanning tree of the dense symmetric weight matrix W: rows (a, b, weight), n-1 of them"""
    n = W.shape[0]
    in_tree = np.zeros(n, dtype=bool)
    best = np.full(n, np.inf)
    src = np.zeros(n, dtype=np.int64)
    edges = np.zeros((n - 1, 3))
    cur = 0
    in_tree[0] = True
    for i in range(n - 1):
        closer = ~in_tree & (W[cur] < best)
        best[closer] = W[cur][closer]
        src[closer] = cur
        cand = np.where(in_tree, np.inf, best)
        nxt = int(np.argmin(cand))
        edges[i] = (src[nxt], nxt, cand[nxt])
        in_tree[nxt] = True
        cur = nxt
    return edges


def _single_linkage(edges: np.ndarray, n: int):
    """sorted MST edges -> dendrogram rows (left, right, distance, size); node ids: points 0..n-1, merges n..2n-2"""
    order = np.argsort(edges[:, 2], kind="stable")
    parent = np.arange(2 * n - 1)
    size = np.ones(2 * n - 1, dtype=np.int64)

    def find(x):
        root = x
        while parent[root] != root:
            root = parent[root]
        while parent[x] != root:
            parent[x], x = root, parent[x]
        return root
    rows = []
    nxt = n
    for a, b, w in edges[order]:
        ra, rb = find(int(a)), find(int(b))
        rows.append((ra, rb, float(w), int(size[ra] + size[rb])))
        parent[ra] = parent[rb] = nxt
        size[nxt] = size[ra] + size[rb]
        nxt += 1
    return rows


def _bfs(rows, n, root):
    out, frontier = [], [root]
    while frontier:
        out.extend(frontier)
        nxt = []
        for x in frontier:
            if x >= n:
                nxt.extend(rows[x - n][:2])
        frontier = nxt
    return out


def hdbscan_labels(X, min_cluster_size: int = 2, min_samples: int | None = None) -> np.ndarray:
    """labels [n] int: 0..k-1 per cluster, -1 = noise (hdbscan.HDBSCAN(min_cluster_size, metric="euclidean").fit_predict)"""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    if min_samples is None:
        min_samples = min_cluster_size
    if n <= max(min_cluster_size, 1):
        return np.full(n, -1, dtype=np.int64)
    sq = (X * X).sum(axis=1)
    D = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (X @ X.T), 0.0))
    np.fill_diagonal(D, 0.0)
    k = min(n - 1, max(min_samples - 1, 0))           # min_samples neighbours INCLUDING the point itself
    core = np.sort(D, axis=1)[:, k]
    W = np.maximum(D, np.maximum(core[:, None], core[None, :]))
    rows = _single_linkage(_mst_prim(W), n)
    # ---- condensed tree: (parent cluster, child, lambda, child size); clusters are numbered from n in BFS order
    root = 2 * n - 2
    relabel = {root: n}
    next_label = n + 1
    ignore = set()
    ctree = []
    for node in _bfs(rows, n, root):
        if node < n or node in ignore:
            continue
        left, right, dist, _ = rows[node - n]
        lam = 1.0 / dist if dist > 0.0 else np.inf
        lc = rows[left - n][3] if left >= n else 1
        rc = rows[right - n][3] if right >= n else 1
        big_l, big_r = lc >= min_cluster_size, rc >= min_cluster_size
        if big_l and big_r:
            for child, cnt in ((left, lc), (right, rc)):
                relabel[child] = next_label
                ctree.append((relabel[node], next_label, lam, cnt))
                next_label += 1
            continue
        for child, big in ((left, big_l), (right, big_r)):
            if big:
                relabel[child] = relabel[node]          # the cluster lives on in its large side
            else:
                for sub in _bfs(rows, n, child):
                    if sub < n:
                        ctree.append((relabel[node], sub, lam, 1))
                    ignore.add(sub)
    # ---- stability and excess-of-mass selection
    birth = {n: 0.0}
    for p, c, lam, cnt in ctree:
        if c >= n:
            birth[c] = lam
    stability = {c: 0.0 for c in birth}
    for p, c, lam, cnt in ctree:
        stability[p] += (lam - birth[p]) * cnt
    children = {c: [] for c in birth}
    for p, c, lam, cnt in ctree:
        if c >= n:
            children[p].append(c)
    selected = {c: True for c in birth}
    for c in sorted(birth, reverse=True):
        if c == n:                                     # the root is never a cluster (allow_single_cluster=False)
            selected[c] = False
            continue
        sub = sum(stability[ch] for ch in children[c])
        if sub > stability[c]:
            selected[c] = False
            stability[c] = sub
        else:
            stack = list(children[c])
            while stack:
                d = stack.pop()
                selected[d] = False
                stack.extend(children[d])
    clusters = sorted(c for c, s in selected.items() if s)
    label_of = {c: i for i, c in enumerate(clusters)}
    # ---- labels: a point belongs to the selected cluster at or above the cluster it fell out of
    up = {c: p for p, c, lam, cnt in ctree if c >= n}
    labels = np.full(n, -1, dtype=np.int64)
    for p, c, lam, cnt in ctree:
        if c < n:
            q = p
            while q not in label_of and q in up:
                q = up[q]
            if q in label_of:
                labels[c] = label_of[q]
    return labels
