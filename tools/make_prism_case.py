"""Writes a prism-mesh case (every hex of a perturbed box split in two prisms) in the polyMesh on-disk format:
python tools/make_prism_case.py <caseDir> [nx ny nz].  Input for tools/solve_case.py when no real case is at hand."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import conftest  # noqa: F401,E402
import fv_case  # noqa: E402

nx, ny, nz = (int(v) for v in sys.argv[2:5]) if len(sys.argv) >= 5 else (40, 40, 30)
m = fv_case.prism_box_mesh(nx, ny, nz, seed=3)
fv_case.write_case(sys.argv[1], m)
print("cells", m["nCells"], "faces", len(m["faces"]))
