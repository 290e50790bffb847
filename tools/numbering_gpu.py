"""TOOL (GPU): the bench's GAMG solve on one stored motorBike mesh under several cell numberings
(tools/numbering_probe.py makes them) - ms per solve to the SAME tolerance, V-cycles per solve, ms per V-cycle, first-solve
time, the engine every level landed on, and (PARITY=1, meshes the oracle solves in seconds) the GaussSeidel sweeps bit for
bit and the residual history against the CPU oracle ON THAT NUMBERING.

    python tools/numbering_gpu.py MESH ORDERING [ORDERING ...]      (env: REPS=5, PARITY=0/1, LEVELS=1 per-level engine table)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numbering_probe as npb  # noqa: E402  (loads the package, builds the DAG helper)
from openfoam_amd import capi, motorbike  # noqa: E402
import torch  # noqa: E402

GAMG = npb.GAMG


def main():
    mesh = sys.argv[1]
    specs = sys.argv[2:]
    reps = int(os.environ.get("REPS", "5"))
    parity = os.environ.get("PARITY", "0") == "1"
    p = motorbike.problem(mesh)
    p.pop("cellLevel"); p.pop("meta")
    nC = p["nCells"]
    rcm = capi.band_compression(nC, p["lowerAddr"], p["upperAddr"])
    dev = torch.device("cuda:0")
    out = {}
    for spec in specs:
        t0 = time.perf_counter()
        order = npb.make_order(p, spec, rcm)
        q = npb.renumber(p, order)
        t_order = time.perf_counter() - t0
        ctx = capi.Context(0)
        t0 = time.perf_counter()
        addr = capi.Addressing(ctx, q["nCells"], q["lowerAddr"], q["upperAddr"], q.get("faceWeights"))
        mat = capi.Matrix(addr)
        d_diag = torch.from_numpy(q["diag"]).to(dev)
        d_upper = torch.from_numpy(q["upper"]).to(dev)
        d_source = torch.from_numpy(q["source"]).to(dev)
        d_psi = torch.zeros(nC, dtype=torch.float64, device=dev)

        def step():
            d_psi.zero_()
            torch.cuda.synchronize()
            mat.set_coeffs(d_diag, d_upper)
            _, perf = mat.solve(d_psi, d_source, history=True, **GAMG)
            torch.cuda.synchronize(); ctx.sync()
            return perf
        perf = step()
        t_first = time.perf_counter() - t0
        mat.wait_plans()
        step()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            perf = step()
            ts.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(ts))
        nv = int(perf["nIterations"])
        r = dict(ms_per_solve=round(ms, 3), vcycles=nv, ms_per_vcycle=round(ms / nv, 3), vcycles_per_s=round(nv / (ms * 1e-3), 2),
                 first_solve_s=round(t_first, 2), ordering_s=round(t_order, 2), fallbacks=int(ctx.fallback_count()),
                 history=[float(h) for h in perf["history"]])
        lv = mat.gamg_level_sizes(**GAMG)
        r["finest"] = dict(levels=int(addr.info()["nLevels"]), engine=addr.sweep_engine(2))
        r["levels"] = [(L["nCells"], L["nLevels"], L["engine_gs_multi"]) for L in lv]
        if parity:
            import oracle_py
            S = oracle_py.System(q)
            rng = np.random.RandomState(5)
            x, b = rng.randn(nC), rng.randn(nC)
            ok = all(np.array_equal(mat.smooth("GaussSeidel", x, b, k), S.smooth("GaussSeidel", x, b, k)) for k in (1, 2, 4))
            ok = ok and np.array_equal(mat.Amul(x), S.Amul(x))
            xo, po = S.solve(q["psi"], q["source"], **GAMG)
            r["parity"] = dict(gs_amul_bitexact=bool(ok), vcycles_oracle=int(po["nIterations"]),
                               history_max_rel=float(np.max(np.abs(np.array(perf["history"]) - np.array(po["history"])) /
                                                            np.abs(np.array(po["history"])))) if len(po["history"]) == len(perf["history"]) else None)
        out[spec] = r
        print("== %-12s %8.2f ms per solve | %d V-cycles | %7.2f ms per V-cycle | %7.1f V-cycles/s | first solve %.2f s | finest %d levels on %s%s"
              % (spec, ms, nv, ms / nv, nv / (ms * 1e-3), t_first, r["finest"]["levels"], r["finest"]["engine"],
                 (" | parity %s" % json.dumps(r["parity"])) if parity else ""), flush=True)
        if os.environ.get("LEVELS", "1") == "1":
            print("   " + " ".join("%d/%d/%s" % t for t in r["levels"]), flush=True)
        mat.close(); addr.close(); ctx.close()
        del d_diag, d_upper, d_source, d_psi
        torch.cuda.empty_cache()
    if os.environ.get("OUT"):
        with open(os.environ["OUT"], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
