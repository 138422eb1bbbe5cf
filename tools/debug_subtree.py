import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from cudapathtracer_amd import api
from oracle import oracle_py as O
from bvh_cases import arrays_of
rng = np.random.default_rng(5)
for n in (6, 10, 20, 40, 64):
    tris = ((rng.random((n, 1, 3)) - 0.5) * 4 + (rng.random((n, 3, 3)) - 0.5) * 0.3).astype(np.float32)
    pts, mesh = arrays_of(tris)
    for leaf in (1, 2):
        on, oi, ost = O.build_bvh(pts, mesh, leaf)
        dn, di, dst = api.build_bvh(pts, mesh, leaf, where="device")
        same = np.array_equal(dn, on) and np.array_equal(di, oi)
        print("n", n, "leaf", leaf, "same", same, "nodes", ost["n_nodes"], dst["n_nodes"], "backups", ost["backups"], dst["backups"])
        if not same:
            a = on.view(np.int32).reshape(-1, 12); b = dn.view(np.int32).reshape(-1, 12)
            print(" oracle idx", oi.tolist()); print(" device idx", di.tolist())
            m = min(len(a), len(b))
            for k in range(m):
                if not np.array_equal(a[k], b[k]):
                    print(" first differing node", k, "oracle", on.view(np.float32).reshape(-1,12)[k,:8].tolist(), a[k,8:].tolist(), "device", dn.view(np.float32).reshape(-1,12)[k,:8].tolist(), b[k,8:].tolist()); break
            print(' stats', ost, dst)
            sub = tris[[3, 4, 0]]
            p2, m2 = arrays_of(sub)
            o2 = O.build_bvh(p2, m2, leaf); d2 = api.build_bvh(p2, m2, leaf, where='device')
            print(' subset [3,4,0] alone: oracle idx', o2[1].tolist(), o2[2], 'device idx', d2[1].tolist(), d2[2])
            sys.exit(0)
