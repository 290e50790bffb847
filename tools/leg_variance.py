"""why the PCG + DIC leg of the bench line ran at 541 it/s (DIC half sweep 0.60 ms) on the driver's box and at
724 it/s (0.44 ms) on ours.  One lease: the leg of bench.py (39 PCG/DIC iterations on the 216^3 box, timed with the factor
computation inside, as bench.py times it) back to back, after idle gaps of 0.5 / 2 / 5 s, and right after a burst of
PCIe traffic (what the host-pointer leg in front of it does); per run the it/s, the mean DIC half-sweep time from HIP
events, the clocks rocm-smi reports before the run.  Prints one JSON line; run it on several leases and keep the table
(profiles/r04_pcg_variance.md)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from openfoam_amd import capi, cases
import numpy as np
import torch


def clocks():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        j = json.loads(r.stdout)
        c = j[sorted(j)[0]]
        return {k.split()[0]: v for k, v in c.items() if "sclk" in k or "mclk" in k or "fclk" in k}
    except Exception as e:
        return dict(error=str(e)[:80])


p = cases.box3d(216)
dev = torch.device("cuda", 0)
ctx = capi.Context(0)
a, m = capi.from_problem(ctx, p)
d_src = torch.from_numpy(p["source"]).to(dev)
d_diag, d_upper = torch.from_numpy(p["diag"]).to(dev), torch.from_numpy(p["upper"]).to(dev)
d_psi = torch.zeros(p["nCells"], dtype=torch.float64, device=dev)
kw = dict(history=False, solver="PCG", preconditioner="DIC", tolerance=0.0, relTol=0.0, maxIter=39)


def leg(fresh_coeffs):
    if fresh_coeffs:
        m.set_coeffs(d_diag, d_upper)     # the factors are recomputed inside the timed solve, as in bench.py
    d_psi.zero_(); torch.cuda.synchronize(); ctx.sync()
    m.profile_begin()
    t0 = time.perf_counter()
    _, pp = m.solve(d_psi, d_src, **kw)
    torch.cuda.synchronize(); ctx.sync()
    dt = time.perf_counter() - t0
    pr = m.profile_end()
    return round(pp["nIterations"] / dt, 1), round(pr["tri_sweep"]["ms"] / pr["tri_sweep"]["count"], 4)


out = dict(first_touch=leg(True), clocks_start=clocks(), runs=[])
for label, idle, pcie in (("back to back", 0, False), ("back to back", 0, False), ("0.5 s idle", 0.5, False), ("2 s idle", 2.0, False),
                          ("5 s idle", 5.0, False), ("after 0.7 GB of pageable PCIe traffic", 0, True), ("back to back", 0, False)):
    if idle:
        time.sleep(idle)
    if pcie:
        h = np.zeros(p["nCells"])
        for _ in range(3):
            m.set_coeffs(p["diag"], p["upper"])
            m.solve(h, p["source"], history=False, inplace=True, solver="GAMG", smoother="GaussSeidel", tolerance=1e-7, relTol=0.01,
                    cacheAgglomeration=1, nCellsInCoarsestLevel=10, mergeLevels=1, agglomerator="faceAreaPair")
    ck = clocks() if idle >= 2 else None
    its, dic = leg(True)
    out["runs"].append(dict(state=label, pcg_dic_it_per_s=its, dic_half_sweep_ms=dic, clocks=ck))
print(json.dumps(out))
m.close(); a.close(); ctx.close()
