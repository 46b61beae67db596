"""How many nearest-neighbour searches could be answered from a per-template-point neighbour list?
(research probe for DESIGN section 9, float64 numpy; NOT part of the product or of the test suite)

For a template point p let N_k(p) be its k nearest template points and R_k(p) the distance to the (k+1)-th.  A query q whose
seed (last iteration's neighbour) is p with d(q,p) < R_k(p)/2 has its nearest neighbour in {p} + N_k(p): any other point x has
d(q,x) >= d(p,x) - d(q,p) > d(q,p).  Prints, per iteration class, the fraction of queries that qualify (seed = previous
neighbour), for k = 6, 8, 12, 16, and how many lanes of a 64-query pass would still need the full search."""
import os, sys
import numpy as np
from scipy.spatial import cKDTree
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from oracle import oracle_py as orc
from perception_amd import capi, synth, templates

def rigid(src, dst):
    cs, cd = src.mean(0), dst.mean(0)
    U, S, Vt = np.linalg.svd((src - cs).T @ (dst - cd))
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rm = Vt.T @ D @ U.T
    return Rm, cd - Rm @ cs

tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE).astype(np.float64)
tree = cKDTree(tpl)
KS = (6, 8, 12, 16)
dd, _ = tree.query(tpl, k=max(KS) + 2)
Rk = {k: dd[:, k + 1] for k in KS}          # column 0 is the point itself
tot = {}
for fi in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    prm = capi.default_params(); prm.rgb_offset = 12
    out = orc.process_frame(synth.frame(fi), prm, tpl.astype(np.float32), want_clouds=True)
    obj = out["objects"].astype(np.float64); lab = out["labels"]
    for c in range(out["result"].n_clusters):
        X = obj[lab == c].copy()
        if len(X) < 100: continue
        nn = None
        for it in range(80):
            d, idx = tree.query(X, k=1)
            if nn is not None:
                dseed = np.linalg.norm(X - tpl[nn], axis=1)
                cls = 1 if it < 16 else 2
                for k in KS:
                    ok = dseed * (1 + 1e-5) < 0.5 * Rk[k][nn]
                    a = tot.setdefault((cls, k), [0, 0, 0, 0]); a[0] += len(X); a[1] += int(ok.sum())
                    npass = (len(X) + 63) // 64
                    pad = np.r_[ok, np.ones(npass * 64 - len(X), bool)].reshape(npass, 64)
                    a[2] += npass; a[3] += int((~pad).sum(1).max() if npass else 0) * 0 + int((~pad).sum())
            nn = idx
            Rm, t = rigid(X, tpl[idx])
            X = X @ Rm.T + t
for (cls, k), (q, s, npass, rest) in sorted(tot.items()):
    print("%-13s k = %2d: %.3f of the queries answered from the neighbour list, %.1f lanes per 64-query pass left for the full search" % (("", "1 <= it < 16", "it >= 16")[cls], k, s / max(q, 1), rest / max(npass, 1)))
