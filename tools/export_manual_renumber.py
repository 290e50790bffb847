"""Writes the cell numbering of ldu_band_compression + ldu_tile_shuffle as the `newToOld` list a `manualRenumber` dictionary reads
(src/renumber/renumberMethods/manualRenumber/manualRenumber.C:60-136: labelIOList <dataFile> in the mesh's facesInstance):

    python tools/export_manual_renumber.py CASE [--tile-size 2048] [--seed 1] [--name cellMap]

then, in the case,  system/renumberMeshDict:  method manual;  manualCoeffs { dataFile "cellMap"; }   and `renumberMesh -overwrite`.
--tile-size 0 writes plain bandCompression (what renumberMesh does by default).  CPU only (host functions of libldugpu.so)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
from openfoam_amd import capi, polymesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--tile-size", type=int, default=2048)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--name", default="cellMap")
    a = ap.parse_args()
    m = polymesh.read_polymesh(a.case)
    l, u = polymesh.ldu_addressing(m)
    order = capi.band_compression(m["nCells"], l, u)
    if a.tile_size > 0:
        order = capi.tile_shuffle(order, a.tile_size, a.seed)
    assert np.array_equal(np.sort(order), np.arange(m["nCells"]))
    path = os.path.join(a.case, "constant", "polyMesh", a.name)
    with open(path, "w") as f:
        f.write("FoamFile\n{\n    version     2.0;\n    format      ascii;\n    class       labelList;\n    location    \"constant/polyMesh\";\n"
                "    object      %s;\n}\n\n%d\n(\n" % (a.name, order.size))
        f.write("\n".join(str(int(v)) for v in order))
        f.write("\n)\n")
    print("wrote %s: %d cells, bandCompression%s" % (path, order.size, (" + tiles of %d (seed %d)" % (a.tile_size, a.seed)) if a.tile_size > 0 else ""))


if __name__ == "__main__":
    main()
