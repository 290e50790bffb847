"""Generates tests/golden/damBreak_2268.npz and tests/golden/interfoam_dambreak*.json - BASELINE config C5's application
(interFoam, damBreak) - with the REFERENCE's own tools, here (the reference does not travel):

  * mesh: the tutorial's blockMeshDict (read from /root/reference at generation time, not stored) through oracle/_ref/blockMesh
    (the reference's blockMeshApp.C, oracle/build_ref_mesh.sh): 2 268 cells;
  * 0/alpha1: the tutorial's setFieldsDict through oracle/_ref/setFields (the reference's setFields.C,
    oracle/build_ref_interfoam.sh);
  * the stock run: oracle/_ref/interFoam (the reference's interFoam.C, unchanged) to t = 0.1 with the tutorial's PCG/DIC for
    p_rgh and with GAMG (BASELINE's wording) - every `Solving for` line, committed as the fixture the plug-in run is held to.

    python tests/golden/make_dambreak_golden.py"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import cavity_case as cc  # noqa: E402
import dambreak_case as dc  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from openfoam_amd import polymesh  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
TUT = "/root/reference/tutorials/multiphase/interFoam/laminar/damBreak"
END = 0.1


def main():
    d = tempfile.mkdtemp()
    case = os.path.join(d, "damBreak")
    os.makedirs(os.path.join(case, "constant", "polyMesh"))
    shutil.copy(os.path.join(TUT, "constant", "polyMesh", "blockMeshDict"), os.path.join(case, "constant", "polyMesh"))
    field = dc.write_dictionaries(case, END)
    shutil.copy(os.path.join(TUT, "system", "setFieldsDict"), os.path.join(case, "system"))
    for app in ("blockMesh", "setFields"):
        if app == "setFields":
            field("alpha1", "volScalarField", "[0 0 0 0 0 0 0]", "uniform 0", "type zeroGradient;",
                  "type inletOutlet; inletValue uniform 0; value uniform 0;")
        r = subprocess.run([os.path.join(REF, app), "-case", case], env=dc.env(), capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(app + " failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    m = polymesh.read_polymesh(case)
    txt = open(os.path.join(case, "0", "alpha1")).read()
    body = txt[txt.index("nonuniform List<scalar>"):]
    n = int(body.split()[2].split("(")[0]) if "(" in body.split()[2] else int(body.split()[2])
    vals = body[body.index("(") + 1:body.index(")")].split()
    alpha = np.array(vals, dtype=np.float64)
    assert alpha.size == m["nCells"] == n == 2268, (alpha.size, m["nCells"], n)
    out = os.path.join(ROOT, "tests", "golden", "damBreak_2268.npz")
    np.savez_compressed(out, points=m["points"], faceStart=m["faceStart"], facePoints=m["facePoints"], owner=m["owner"],
                        neighbour=m["neighbour"], patchNames=np.array([p["name"] for p in m["patches"]]),
                        patchTypes=np.array([p["type"] for p in m["patches"]]),
                        patchSize=np.array([p["nFaces"] for p in m["patches"]]),
                        patchStart=np.array([p["startFace"] for p in m["patches"]]), alpha1=alpha)
    print("wrote", out, m["nCells"], "cells,", int(alpha.sum()), "cells of water")
    shutil.rmtree(d)
    for tag, ps in (("", None), ("_gamg", dc.GAMG)):
        d = tempfile.mkdtemp()
        case = os.path.join(d, "damBreak")
        dc.write(case, END, p_solver=ps)
        lines = cc.solve_lines(dc.run(case))
        json.dump(dict(end_time=END, lines=lines), open(os.path.join(ROOT, "tests", "golden", "interfoam_dambreak%s.json" % tag), "w"))
        print(tag or "pcg", len(lines), "solver lines;", lines[:3], lines[-1])
        shutil.rmtree(d)


if __name__ == "__main__":
    main()
