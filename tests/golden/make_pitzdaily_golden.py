"""Generates tests/golden/pitzDaily_*.npz - config C2 of BASELINE.json (simpleFoam pitzDaily, ~12k cells, GAMG p-solve):

  * the mesh: the tutorial's blockMeshDict (read from /root/reference at generation time, not stored) run through the
    REFERENCE's own blockMesh library (oracle/_ref/blockmesh_driver, src/mesh/blockMesh compiled by
    oracle/build_ref_fv.sh): 12 225 cells, 24 170 internal faces - SURVEY.md 8's numbers;
  * the pressure equation of pEqn.H on it, assembled by the reference's own gaussLaplacianScheme / correctedSnGrad with
    the tutorial's fvSchemes and 0/p boundary conditions (oracle/fv_driver.C mode pitz) and solved by the reference's
    own fvScalarMatrix::solve with the GAMG block of motorBike/system/fvSolution:19-31 and with
    the tutorial's own PCG-free alternative (PCG/DIC) - residual histories, psi, flux.

Run here (the reference does not travel):   python tests/golden/make_pitzdaily_golden.py"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import fv_case  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from openfoam_amd import polymesh  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
TUTORIAL = "/root/reference/tutorials/incompressible/simpleFoam/pitzDaily"
# motorBike/system/fvSolution:19-31 (the block SURVEY.md 8d prescribes for C2); agglomeration cache irrelevant for one solve
GAMG = ("solver GAMG; tolerance 1e-7; relTol 0.01; smoother GaussSeidel; nPreSweeps 0; nPostSweeps 2; "
        "cacheAgglomeration on; agglomerator faceAreaPair; nCellsInCoarsestLevel 10; mergeLevels 1;")
GAMG_TIGHT = GAMG.replace("relTol 0.01", "relTol 0").replace("tolerance 1e-7", "tolerance 1e-9")
PCG = "solver PCG; preconditioner DIC; tolerance 1e-7; relTol 0.01;"
SCHEMES = ("ddtSchemes { default steadyState; }\ngradSchemes { default Gauss linear; }\n"
           "divSchemes { default Gauss linear; }\nlaplacianSchemes { default Gauss linear corrected; }\n"
           "interpolationSchemes { default linear; }\nsnGradSchemes { default corrected; }\n"
           "fluxRequired { default no; p; }\n")


def env():
    return dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x", WM_PROJECT_DIR=REF,
                LD_LIBRARY_PATH=REF + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FOAM_SIGFPE="false")


def make_case(d):
    case = os.path.join(d, "case")
    os.makedirs(os.path.join(case, "constant", "polyMesh"))
    os.makedirs(os.path.join(case, "system"))
    shutil.copy(os.path.join(TUTORIAL, "constant", "polyMesh", "blockMeshDict"), os.path.join(case, "constant", "polyMesh"))
    with open(os.path.join(case, "system", "controlDict"), "w") as f:
        f.write(fv_case.HEADER % ("dictionary", "system", "controlDict"))
        f.write("application simpleFoam;\nstartFrom startTime;\nstartTime 0;\nstopAt endTime;\nendTime 1;\ndeltaT 1;\n"
                "writeControl timeStep;\nwriteInterval 1;\nwriteFormat ascii;\nwritePrecision 6;\ntimeFormat general;\n"
                "timePrecision 6;\nrunTimeModifiable false;\n")
    with open(os.path.join(case, "system", "fvSchemes"), "w") as f:
        f.write(fv_case.HEADER % ("dictionary", "system", "fvSchemes"))
        f.write(SCHEMES)
    with open(os.path.join(case, "system", "fvSolution"), "w") as f:
        f.write(fv_case.HEADER % ("dictionary", "system", "fvSolution"))
        f.write("solvers { }\n")
    r = subprocess.run([os.path.join(REF, "blockmesh_driver"), case], env=env(), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("blockmesh_driver failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    return case


def run_pitz(case, solver_dict):
    outp = os.path.join(case, "out.bin")
    r = subprocess.run([os.path.join(REF, "fv_driver"), case, "-", outp, "pitz", solver_dict], env=env(),
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("fv_driver pitz failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    res = {}
    with open(outp, "rb") as f:
        while True:
            hdr = f.read(64)
            if len(hdr) < 64:
                break
            name, n = hdr.split(b"\0", 1)[0].decode().split()
            res[name] = np.fromfile(f, dtype=np.float64, count=int(n))
    # per-iteration residuals of the OUTER solver only (the coarsest-level DICPCG of GAMG prints the same kind of line)
    name = "GAMG" if "solver GAMG" in solver_dict else "DICPCG"
    hist = np.array([float(m.group(2)) for m in re.finditer(r"^%s:  Iteration (\d+) residual = (\S+)" % name, r.stdout, re.M)])
    return res, hist


def generate():
    with tempfile.TemporaryDirectory() as d:
        case = make_case(d)
        mesh = polymesh.read_polymesh(case)
        res, hist = run_pitz(case, GAMG)
        res2, hist2 = run_pitz(case, GAMG_TIGHT)
        res3, hist3 = run_pitz(case, PCG)
    out = dict(points=mesh["points"], faceStart=mesh["faceStart"], facePoints=mesh["facePoints"], owner=mesh["owner"],
               neighbour=mesh["neighbour"], nCells=np.array([mesh["nCells"]], dtype=np.int32),
               patchNames=np.array([p["name"] for p in mesh["patches"]]),
               patchTypes=np.array([p["type"] for p in mesh["patches"]]),
               patchStart=np.array([p["startFace"] for p in mesh["patches"]], dtype=np.int32),
               patchSize=np.array([p["nFaces"] for p in mesh["patches"]], dtype=np.int32),
               gamg_dict=np.array(GAMG), gamg_tight_dict=np.array(GAMG_TIGHT), pcg_dict=np.array(PCG))
    # geometry is not stored: the tests rebuild it from the points on the device and are held to the reference's
    # MATRIX (upper / diag / source bit for bit), which they can only reproduce with bit-exact geometry
    skip = ("V", "Sf", "magSf", "C", "weights", "nonOrthDeltaCoeffs")
    for k, v in res.items():
        if k in skip:
            continue
        if k.endswith("_faceCells"):
            v = v.astype(np.int32)
        out[k if k.startswith(("ref_", "p0_", "p1_", "p2_", "p3_", "p4_")) else "ref_" + k] = v
    out["ref_nonOrthCorrectionVectors_max"] = np.array([np.abs(res["nonOrthCorrectionVectors"]).max()])
    del out["ref_nonOrthCorrectionVectors"]
    out["ref_gamg_history"] = hist
    out["ref_gamg_tight_history"], out["ref_gamg_tight_perf"], out["ref_gamg_tight_psi"] = hist2, res2["ref_perf"], res2["ref_psi"]
    out["ref_pcg_history"], out["ref_pcg_perf"] = hist3, res3["ref_perf"]
    return out


if __name__ == "__main__":
    if not os.path.exists(os.path.join(REF, "blockmesh_driver")) or not fv_case.driver_available():
        raise SystemExit("oracle/_ref/{blockmesh_driver,fv_driver} missing: run oracle/build_ref_fv.sh (needs /root/reference)")
    data = generate()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pitzDaily_12225.npz")
    np.savez_compressed(path, **data)
    print("pitzDaily: cells", int(data["nCells"][0]), "internal faces", data["neighbour"].size, "patches",
          list(zip(data["patchNames"], data["patchTypes"], data["patchSize"])))
    print("GAMG (motorBike block):", data["ref_perf"], "history", data["ref_gamg_history"])
    print("GAMG tight:", data["ref_gamg_tight_perf"], len(data["ref_gamg_tight_history"]), "residuals")
    print("PCG/DIC:", data["ref_pcg_perf"])
    print("max |nonOrthCorrectionVectors|", data["ref_nonOrthCorrectionVectors_max"], "file", os.path.getsize(path) // 1024, "KiB")
