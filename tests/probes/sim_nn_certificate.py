"""How often would a triangle-inequality certificate let an ICP iteration skip a query's nearest-neighbour search?
(research probe for DESIGN section 9, float64 numpy; NOT part of the product or of the test suite)

After a full search at iteration k a query knows d1 (distance to its neighbour p) and a lower bound L2 of the distance to
every other template point.  While  d1' + slack < L2 - M  (d1' = current distance to p, M = the query's accumulated motion since
the search) p is still the unique nearest neighbour and the search can be skipped.  Prints, per iteration class and for far
(d1 > one grid cell) / near queries, the fraction of searches the certificate would skip when L2 is the true second-nearest
distance (the best any implementation could know)."""
import os, sys
import numpy as np
from scipy.spatial import cKDTree
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from oracle import oracle_py as orc
from perception_amd import capi, synth, templates

def rigid(src, dst):
    cs, cd = src.mean(0), dst.mean(0)
    H = (src - cs).T @ (dst - cd)
    U, S, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    Rm = Vt.T @ D @ U.T
    return Rm, cd - Rm @ cs

tpl = templates.template_xyz32(**templates.DEFAULT_TEMPLATE).astype(np.float64)
tree = cKDTree(tpl)
CELL = 0.00504
tot = {}
for fi in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    prm = capi.default_params(); prm.rgb_offset = 12
    out = orc.process_frame(synth.frame(fi), prm, tpl.astype(np.float32), want_clouds=True)
    obj = out["objects"].astype(np.float64)
    lab = out["labels"]
    for k in range(out["result"].n_clusters):
        X = obj[lab == k].copy()
        if len(X) < 100: continue
        n = len(X)
        nn = np.zeros(n, int); L2 = np.full(n, -1.0); M = np.zeros(n)
        for it in range(80):
            d, idx = tree.query(X, k=2)
            d1_now = np.linalg.norm(X - tpl[nn], axis=1)
            ok = (it > 0) & (d1_now + 1e-6 * (1 + d1_now) < L2 - M)      # certificate from the last full search
            assert np.all(idx[ok, 0] == nn[ok])                           # (it is exact)
            far = d[:, 0] > CELL
            cls = 0 if it < 3 else 1 if it < 16 else 2
            for nm, sel in (("far", far), ("near", ~far)):
                a = tot.setdefault((cls, nm), [0, 0]); a[0] += int(sel.sum()); a[1] += int((ok & sel).sum())
            # searched queries refresh their certificate
            s = ~ok
            nn[s] = idx[s, 0]; L2[s] = d[s, 1]; M[s] = 0.0
            Rm, t = rigid(X, tpl[idx[:, 0]])
            Xn = X @ Rm.T + t
            M += np.linalg.norm(Xn - X, axis=1)
            X = Xn
for (cls, nm), (q, s) in sorted(tot.items()):
    print("%-13s %-4s queries %8d  skippable %.3f" % (("it < 3", "3 <= it < 16", "it >= 16")[cls], nm, q, s / max(q, 1)))
