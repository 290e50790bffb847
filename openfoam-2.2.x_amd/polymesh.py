"""Reader for the polyMesh on-disk format of OpenFOAM-2.2.x (constant/polyMesh/{points,faces,owner,neighbour,
boundary}, ascii and binary) - the data format on the mesh side of the path: what
bench.py --mesh motorbike, tools/solve_case.py and the tests use to put a real case's matrix addressing and geometry in
front of the solvers.

Format (src/OpenFOAM/meshes/polyMesh/polyMeshFromShapeMesh.C / polyMeshIO.C write these with the standard List
and dictionary writers): a FoamFile header dictionary, then `N ( item ... )`; points items are `(x y z)`, faces
items `n(p0 p1 ...)`, owner / neighbour items labels, boundary items `name { type T; nFaces n; startFace s; ... }`.
Comments are C / C++ style.  `writeFormat binary` (IOstream::BINARY: the list body is the raw little-endian array between
`N\n(` and `)`, label = int32, scalar = float64 in this build - UList<T>::writeEntry / List<T>::readList, UListIO.C:74-96,
ListIO.C:70-150) is read for labelList / vectorField, and faces come as the faceCompactList the binary writer uses
(polyMeshIO / faceIOList: an offset list of nFaces+1 labels followed by the flat list of point labels,
CompactListList IO).  Compressed (.gz) files and ascii faceCompactList are refused with a clear error."""
import os
import re

import numpy as np

_COMMENT = re.compile(r"/\*.*?\*/|//[^\n]*", re.S)


def _header(raw, path):
    """FoamFile header dictionary of a file and the byte offset behind it"""
    m = re.search(rb"FoamFile\s*\{(.*?)\}", raw[:4096], re.S)
    header = {}
    end = 0
    if m:
        for k, v in re.findall(r"(\w+)\s+([^;]+);", m.group(1).decode("ascii", "replace")):
            header[k] = v.strip()
        end = m.end()
    return header, end


def _binary_list(raw, pos, dtype, width, path):
    """one binary list `N ( raw bytes )` starting at or after byte pos -> (array[N, width], position behind it)"""
    m = re.compile(rb"(\d+)\s*\(").search(raw, pos)
    if not m:
        raise ValueError("%s: no `N (` list found" % path)
    n = int(m.group(1))
    nbytes = n * width * np.dtype(dtype).itemsize
    a = np.frombuffer(raw, dtype=dtype, count=n * width, offset=m.end())
    if raw[m.end() + nbytes:m.end() + nbytes + 1] != b")":
        raise ValueError("%s: binary list of %d items is not closed where expected (label / scalar width?)" % (path, n))
    return a.reshape(n, width) if width > 1 else a, m.end() + nbytes + 1


def _skip_comments(raw, pos):
    """position of the first list size behind the `// * * *` banner line that follows the header"""
    m = re.compile(rb"(?://[^\n]*\n|\s)*").match(raw, pos)
    return m.end() if m else pos


def _read_raw(path):
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        raise ValueError("%s.gz: compressed polyMesh files are not supported (writeCompression off)" % path)
    with open(path, "rb") as f:
        return f.read()


def _body(path):
    raw = _read_raw(path)
    try:
        text = raw.decode("ascii")
    except UnicodeDecodeError:
        raise ValueError("%s: not an ascii polyMesh file" % path)
    text = _COMMENT.sub(" ", text)
    m = re.search(r"FoamFile\s*\{(.*?)\}", text, re.S)
    header = {}
    if m:
        for k, v in re.findall(r"(\w+)\s+([^;]+);", m.group(1)):
            header[k] = v.strip()
        text = text[m.end():]
    if header.get("format", "ascii") != "ascii" and header.get("class") != "polyBoundaryMesh":
        raise ValueError("%s: format %s where ascii was expected" % (path, header["format"]))
    return header, text


def _list_payload(text, path):
    m = re.search(r"(\d+)\s*\(", text)
    if not m:
        raise ValueError("%s: no `N (` list found" % path)
    n = int(m.group(1))
    end = text.rfind(")")
    return n, text[m.end():end]


def read_points(path):
    raw = _read_raw(path)
    header, end = _header(raw, path)
    if header.get("format") == "binary":
        pts, _ = _binary_list(raw, _skip_comments(raw, end), "<f8", 3, path)
        return pts.copy()
    _, text = _body(path)
    n, body = _list_payload(text, path)
    vals = np.array(body.replace("(", " ").replace(")", " ").split(), dtype=np.float64)
    if vals.size != 3 * n:
        raise ValueError("%s: %d values for %d points" % (path, vals.size, n))
    return vals.reshape(n, 3)


def read_labels(path):
    raw = _read_raw(path)
    header, end = _header(raw, path)
    if header.get("format") == "binary":
        vals, _ = _binary_list(raw, _skip_comments(raw, end), "<i4", 1, path)
        return vals.copy(), header
    header, text = _body(path)
    n, body = _list_payload(text, path)
    vals = np.array(body.split(), dtype=np.int64)
    if vals.size != n:
        raise ValueError("%s: %d labels, header says %d" % (path, vals.size, n))
    return vals.astype(np.int32), header


def read_faces(path):
    """-> (faceStart[nFaces+1], facePoints) CSR of point labels"""
    raw = _read_raw(path)
    header, end = _header(raw, path)
    if header.get("format") == "binary":
        if header.get("class") != "faceCompactList":
            raise ValueError("%s: binary faces of class %s (faceCompactList expected)" % (path, header.get("class")))
        start, pos = _binary_list(raw, _skip_comments(raw, end), "<i4", 1, path)
        pts, _ = _binary_list(raw, pos, "<i4", 1, path)
        if start.size == 0 or start[0] != 0 or start[-1] != pts.size or np.any(np.diff(start) < 3):
            raise ValueError("%s: malformed faceCompactList" % path)
        return start.copy(), pts.copy()
    header, text = _body(path)
    if header.get("class") == "faceCompactList":
        raise ValueError("%s: faceCompactList is not supported (write the mesh uncompacted)" % path)
    n, body = _list_payload(text, path)
    toks = body.replace("(", " ( ").replace(")", " ) ").split()
    start = np.zeros(n + 1, dtype=np.int32)
    pts = []
    i = f = 0
    while i < len(toks):
        k = int(toks[i])
        if toks[i + 1] != "(" or toks[i + 2 + k] != ")":
            raise ValueError("%s: malformed face %d" % (path, f))
        pts.extend(toks[i + 2:i + 2 + k])
        start[f + 1] = start[f] + k
        f += 1
        i += 3 + k
    if f != n:
        raise ValueError("%s: %d faces, header says %d" % (path, f, n))
    return start, np.array(pts, dtype=np.int32)


def read_boundary(path):
    _, text = _body(path)
    n, body = _list_payload(text, path)
    patches = []
    for name, inner in re.findall(r"([^\s{}()]+)\s*\{(.*?)\}", body, re.S):
        d = {k: v.strip() for k, v in re.findall(r"(\w+)\s+([^;]+);", inner)}
        patches.append(dict(name=name, type=d.get("type", "patch"), nFaces=int(d["nFaces"]),
                            startFace=int(d["startFace"]), neighbourPatch=d.get("neighbourPatch"),
                            myProcNo=int(d["myProcNo"]) if "myProcNo" in d else None,
                            neighbProcNo=int(d["neighbProcNo"]) if "neighbProcNo" in d else None))
    if len(patches) != n:
        raise ValueError("%s: %d patches, header says %d" % (path, len(patches), n))
    return patches


def read_polymesh(case_or_dir):
    """-> dict(points, faceStart, facePoints, owner, neighbour, nCells, nInternalFaces, patches) with the checks
    polyMesh itself makes on construction (sizes consistent, patches contiguous behind the internal faces)."""
    d = case_or_dir
    if not os.path.exists(os.path.join(d, "owner")):
        d = os.path.join(case_or_dir, "constant", "polyMesh")
    points = read_points(os.path.join(d, "points"))
    faceStart, facePoints = read_faces(os.path.join(d, "faces"))
    owner, _ = read_labels(os.path.join(d, "owner"))
    neighbour, _ = read_labels(os.path.join(d, "neighbour"))
    patches = read_boundary(os.path.join(d, "boundary"))
    nFaces, nInt = owner.size, neighbour.size
    if faceStart.size - 1 != nFaces:
        raise ValueError("faces (%d) and owner (%d) disagree" % (faceStart.size - 1, nFaces))
    if nInt > nFaces:
        raise ValueError("more neighbours than faces")
    nCells = int(max(owner.max(initial=-1), neighbour.max(initial=-1))) + 1
    nxt = nInt
    for p in patches:
        if p["startFace"] != nxt:
            raise ValueError("patch %s starts at face %d, expected %d" % (p["name"], p["startFace"], nxt))
        nxt += p["nFaces"]
    if nxt != nFaces:
        raise ValueError("patches end at face %d of %d" % (nxt, nFaces))
    if facePoints.size and (facePoints.min() < 0 or facePoints.max() >= points.shape[0]):
        raise ValueError("point label out of range")
    return dict(points=points, faceStart=faceStart, facePoints=facePoints, owner=owner, neighbour=neighbour,
                nCells=nCells, nInternalFaces=nInt, patches=patches)


def ldu_addressing(mesh):
    """lowerAddr / upperAddr of the internal faces (fvMesh::lduAddr: owner / neighbour), checked for the
    upper-triangular order the solvers rely on (lduAddressing.C:92-126)."""
    nI = mesh["nInternalFaces"]
    l, u = mesh["owner"][:nI].astype(np.int32), mesh["neighbour"].astype(np.int32)
    if nI and not np.all(l < u):
        raise ValueError("internal faces are not owner < neighbour")
    key = l.astype(np.int64) * (int(u.max(initial=0)) + 1) + u
    if nI and not np.all(np.diff(key) > 0):
        raise ValueError("internal faces are not in upper-triangular order (run renumberMesh / use "
                         "capi.renumber_addressing)")
    return l, u
