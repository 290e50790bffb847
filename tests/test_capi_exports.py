"""CPU (-m "not gpu"): the drop-in boundary itself.

* libldugpu.so loads here (no GPU) and exports EVERY function include/ldugpu.h declares - the header is what a maintainer binds
  against, a declared entry point that the library does not carry is a link error on their side;
* host-side pieces of the library that need no device: the block engine's footprint-balanced partitioner
  (ldu_partition_blobs_footprint, csrc/ldu_mesh.hip: partition_blobs_slots)."""
import os
import re

import numpy as np
import pytest

from openfoam_amd import capi, cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ldugpu.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)          # comments out
    text = re.sub(r"//[^\n]*", " ", text)
    names = re.findall(r"\b(ldu_[A-Za-z0-9_]+)\s*\(", text)
    # (typedef'd callback types and macros are not functions: only identifiers followed by a parameter list that ends in `);`)
    decl = set()
    for m in re.finditer(r"\b(ldu_[A-Za-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        decl.add(m.group(1))
    return sorted(decl & set(names))


def test_library_exports_every_declared_function():
    lib = capi.lib()
    names = declared_functions()
    assert len(names) > 100, names          # the header really was parsed
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/ldugpu.h but not exported by libldugpu.so: %s" % missing
    # ... and the Python binding's own list is part of the header (nothing bound that the header does not declare)
    extra = [n for n in capi.EXPORTS if n not in names]
    assert not extra, "bound by capi.py but not declared in include/ldugpu.h: %s" % extra


def footprints(nC, l, u, part, nP):
    """cells + distinct outside neighbours of every part"""
    foot = np.bincount(part, minlength=nP).astype(np.int64)
    a = np.r_[l, u].astype(np.int64)
    b = np.r_[u, l].astype(np.int64)
    cut = part[a] != part[b]
    key = np.unique(part[a][cut].astype(np.int64) * nC + b[cut])      # (part, outside cell) pairs
    return foot + np.bincount(key // nC, minlength=nP)


@pytest.mark.parametrize("gen,target", [(lambda: cases.box3d(24), 900), (lambda: cases.irregular_box(20), 700),
                                        (lambda: cases.laplacian2d(60, 45), 300)])
def test_footprint_partition(gen, target):
    p = gen()
    nC, l, u = p["nCells"], p["lowerAddr"], p["upperAddr"]
    part, nP = capi.partition_blobs_footprint(nC, l, u, target, nC)
    assert part.min() == 0 and part.max() == nP - 1 and np.unique(part).size == nP       # every cell in a part, labels dense
    foot = footprints(nC, l, u, part, nP)
    # a blob stops growing before it would pass the target; what joins afterwards are pockets of less than a sixth of a
    # footprint each (the block engine asks for 92 % / 85 % of what fits and checks the real sizes: bk_build)
    assert foot.max() <= 1.35 * target, (foot.max(), target)
    assert np.median(foot) >= 0.6 * target                                              # ... and the blobs are not crumbs
    assert nP <= 2.2 * (foot.sum() / target) + 2
    again, nP2 = capi.partition_blobs_footprint(nC, l, u, target, nC)
    assert nP2 == nP and np.array_equal(again, part)                                     # deterministic
    with pytest.raises(capi.LduError):
        capi.partition_blobs_footprint(nC, l, u, target, max(1, nP // 3))                # more parts than allowed: refused
