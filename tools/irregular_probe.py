"""sweep engines on the unstructured stand-in (cases.irregular_box renumbered by bandCompression): plan statistics
(LDU_VERBOSE=1) and sweep times per engine.   python tools/irregular_probe.py [n=100]"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import torch  # before libldugpu: one HIP runtime per process
torch.cuda.init()
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
p = cases.irregular_box(n)
if os.environ.get("PROBE_RENUMBER", "1") == "1":
    order = capi.band_compression(p["nCells"], p["lowerAddr"], p["upperAddr"])
    nl, nu, fmap, flip = capi.renumber_addressing(p["nCells"], p["lowerAddr"], p["upperAddr"], order)
    p = cases.renumbered(p, order, fmap, flip, nl, nu)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
print("n", n, "info", a.info(), "engines", [a.sweep_engine(k) for k in (0, 1, 2)], flush=True)
rng = np.random.RandomState(1)
x, b = rng.randn(p["nCells"]), rng.randn(p["nCells"])
dx, db = torch.from_numpy(x).cuda(), torch.from_numpy(b).cuda()
ks = [int(v) for v in os.environ.get("PROBE_KS", "1,2,4").split(",")]
ops = [("GS%d" % k, (lambda k=k: capi.lib().ldu_smooth(m.h, 0, capi._ptr(dx), capi._ptr(db), k))) for k in ks]
ops.append(("DIC", lambda: capi.lib().ldu_precondition(m.h, 2, capi._ptr(dx), capi._ptr(db), 0)))
for name, fn in ops:
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    ctx.sync()
    print(name, "%.3f ms" % (1e3 * (time.perf_counter() - t0) / 5), "fallbacks", ctx.fallback_count(), flush=True)
