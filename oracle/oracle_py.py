"""TEST INFRASTRUCTURE - ctypes binding of the CPU oracle (oracle/libldu_oracle.so)
and a runner for the real reference build (oracle/_ref/ref_driver).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SOLVERS = {"PCG": 0, "PBiCG": 1, "smoothSolver": 2, "GAMG": 3, "diagonal": 4}
PRECONDS = {"none": 0, "diagonal": 1, "DIC": 2, "FDIC": 3, "DILU": 4, "GAMG": 5}
SMOOTHERS = {"GaussSeidel": 0, "symGaussSeidel": 1, "DIC": 2, "DILU": 3,
             "DICGaussSeidel": 4, "DILUGaussSeidel": 5, "FDIC": 6, "nonBlockingGaussSeidel": 7}
AGGLOMERATORS = {"faceAreaPair": 0, "algebraicPair": 1}
# coupled family LduMatrix<Type, scalar, scalar> (ldu_oracle_coupled.c)
CSOLVERS = {"PCICG": 0, "PBiCCCG": 1, "PBiCICG": 2, "SmoothSolver": 3, "diagonal": 4}
CPRECONDS = {"none": 0, "diagonal": 1, "DILU": 2}


class COpts(C.Structure):
    _fields_ = [("solver", C.c_int), ("precond", C.c_int), ("smoother", C.c_int), ("nc", C.c_int),
                ("maxIter", C.c_int), ("nSweeps", C.c_int),
                ("tolerance", C.c_double * 9), ("relTol", C.c_double * 9), ("ipw", C.c_double * 9)]


class CPerf(C.Structure):
    _fields_ = [("initialResidual", C.c_double * 9), ("finalResidual", C.c_double * 9),
                ("normFactor", C.c_double * 9), ("singular", C.c_int * 9),
                ("nIterations", C.c_int), ("converged", C.c_int)]


class Patch(C.Structure):
    _fields_ = [("n", C.c_int), ("faceCells", C.c_void_p), ("bouCoeffs", C.c_void_p),
                ("intCoeffs", C.c_void_p), ("nbrDom", C.c_int), ("nbrPatch", C.c_int)]


class Dom(C.Structure):
    _fields_ = [("nCells", C.c_int), ("nFaces", C.c_int), ("l", C.c_void_p), ("u", C.c_void_p),
                ("diag", C.c_void_p), ("upper", C.c_void_p), ("lower", C.c_void_p),
                ("nPatches", C.c_int), ("patches", C.c_void_p), ("cellOffset", C.c_int),
                ("losort", C.c_void_p), ("ownerStart", C.c_void_p), ("losortStart", C.c_void_p)]


class Sys(C.Structure):
    _fields_ = [("nDom", C.c_int), ("dom", C.c_void_p), ("nCellsTotal", C.c_int)]


class Opts(C.Structure):
    _fields_ = [("solver", C.c_int), ("precond", C.c_int), ("smoother", C.c_int),
                ("tolerance", C.c_double), ("relTol", C.c_double), ("maxIter", C.c_int),
                ("nSweeps", C.c_int),
                ("nPreSweeps", C.c_int), ("preSweepsLevelMultiplier", C.c_int), ("maxPreSweeps", C.c_int),
                ("nPostSweeps", C.c_int), ("postSweepsLevelMultiplier", C.c_int), ("maxPostSweeps", C.c_int),
                ("nFinestSweeps", C.c_int), ("interpolateCorrection", C.c_int), ("scaleCorrection", C.c_int),
                ("nCellsInCoarsestLevel", C.c_int), ("mergeLevels", C.c_int), ("agglomerator", C.c_int),
                ("nVcycles", C.c_int), ("directSolveCoarsest", C.c_int)]


class Perf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("normFactor", C.c_double), ("nIterations", C.c_int), ("converged", C.c_int),
                ("singular", C.c_int), ("nHist", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libldu_oracle.so", "libldu_oracle_omp.so"])


def lib():
    """LDU_ORACLE_OMP=1 (set before the first call) selects the OpenMP build: one thread per sub-domain = per
    emulated rank, bit-identical results (oracle/ldu_oracle.h ORC_PAR); OMP_NUM_THREADS sets the cores used."""
    global _LIB
    if _LIB is None:
        omp = os.environ.get("LDU_ORACLE_OMP") == "1"
        path = os.path.join(HERE, "libldu_oracle_omp.so" if omp else "libldu_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_solve.restype = Perf
        L.orc_gamg_build.restype = C.c_void_p
        for nm in ("orc_gamg_restrict", "orc_gamg_faceRestrict", "orc_gamg_level_lower",
                   "orc_gamg_level_upper", "orc_gamg_level_diag", "orc_gamg_level_upperCoeffs",
                   "orc_gamg_level_lowerCoeffs"):
            getattr(L, nm).restype = C.c_void_p
        L.orc_gSumProd.restype = C.c_double
        L.orc_gSumMag.restype = C.c_double
        L.orc_normFactor.restype = C.c_double
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_opts(**kw):
    o = Opts()
    lib().orc_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "solver":
            v = SOLVERS[v]
        elif k in ("precond", "preconditioner"):
            k = "precond"
            v = PRECONDS[v]
        elif k == "smoother":
            v = SMOOTHERS[v]
        elif k == "agglomerator":
            v = AGGLOMERATORS[v]
        setattr(o, k, v)
    return o


class System:
    """A list of sub-domain problems (dicts as made by cases.py, optionally with
    'patches': [dict(faceCells, bouCoeffs, intCoeffs, nbrDom, nbrPatch)])."""

    def __init__(self, problems):
        if isinstance(problems, dict):
            problems = [problems]
        self.problems = problems
        self.keep = []
        n = len(problems)
        self.doms = (Dom * n)()
        for d, p in enumerate(problems):
            a = {}
            a["l"] = np.ascontiguousarray(p["lowerAddr"], dtype=np.int32)
            a["u"] = np.ascontiguousarray(p["upperAddr"], dtype=np.int32)
            a["diag"] = np.ascontiguousarray(p["diag"], dtype=np.float64)
            a["upper"] = np.ascontiguousarray(p["upper"], dtype=np.float64)
            a["lower"] = (np.ascontiguousarray(p["lower"], dtype=np.float64)
                          if "lower" in p else a["upper"])
            self.keep.append(a)
            D = self.doms[d]
            D.nCells = int(p["nCells"])
            D.nFaces = int(a["l"].size)
            D.l, D.u = _p(a["l"]), _p(a["u"])
            D.diag, D.upper, D.lower = _p(a["diag"]), _p(a["upper"]), _p(a["lower"])
            pats = p.get("patches", [])
            D.nPatches = len(pats)
            if pats:
                arr = (Patch * len(pats))()
                for i, q in enumerate(pats):
                    fc = np.ascontiguousarray(q["faceCells"], dtype=np.int32)
                    bc = np.ascontiguousarray(q["bouCoeffs"], dtype=np.float64)
                    ic = np.ascontiguousarray(q["intCoeffs"], dtype=np.float64)
                    self.keep.append((fc, bc, ic))
                    arr[i].n = fc.size
                    arr[i].faceCells, arr[i].bouCoeffs, arr[i].intCoeffs = _p(fc), _p(bc), _p(ic)
                    arr[i].nbrDom, arr[i].nbrPatch = int(q["nbrDom"]), int(q["nbrPatch"])
                self.keep.append(arr)
                D.patches = C.cast(arr, C.c_void_p)
        self.sys = Sys()
        self.sys.nDom = n
        self.sys.dom = C.cast(self.doms, C.c_void_p)
        lib().orc_sys_finalize(C.byref(self.sys))
        self.n = self.sys.nCellsTotal
        self.sym = all("lower" not in p for p in problems)

    def __del__(self):
        try:
            lib().orc_sys_free_derived(C.byref(self.sys))
        except Exception:
            pass

    # --- addressing of domain d
    def addressing(self, d=0):
        D = self.doms[d]
        nF, nC = D.nFaces, D.nCells
        lo = np.ctypeslib.as_array(C.cast(D.losort, C.POINTER(C.c_int)), shape=(max(nF, 1),))[:nF].copy()
        os_ = np.ctypeslib.as_array(C.cast(D.ownerStart, C.POINTER(C.c_int)), shape=(nC + 1,)).copy()
        ls = np.ctypeslib.as_array(C.cast(D.losortStart, C.POINTER(C.c_int)), shape=(nC + 1,)).copy()
        return lo, os_, ls

    def _vec(self, x=None):
        if x is None:
            return np.zeros(self.n)
        return np.ascontiguousarray(x, dtype=np.float64).copy()

    def Amul(self, psi):
        y = self._vec(); x = self._vec(psi)
        lib().orc_Amul(C.byref(self.sys), _p(y), _p(x)); return y

    def Tmul(self, psi):
        y = self._vec(); x = self._vec(psi)
        lib().orc_Tmul(C.byref(self.sys), _p(y), _p(x)); return y

    def sumA(self):
        y = self._vec()
        lib().orc_sumA(C.byref(self.sys), _p(y)); return y

    def residual(self, psi, source):
        y = self._vec(); x = self._vec(psi); b = self._vec(source)
        lib().orc_residual(C.byref(self.sys), _p(y), _p(x), _p(b)); return y

    def dom_op(self, name, *vecs, out_faces=False, d=0):
        D = self.doms[d]
        out = np.zeros(D.nFaces if out_faces else D.nCells)
        args = [_p(np.ascontiguousarray(v, dtype=np.float64)) for v in vecs]
        getattr(lib(), name)(C.byref(D), _p(out), *args)
        return out

    def gSumProd(self, a, b):
        return lib().orc_gSumProd(C.byref(self.sys), _p(self._vec(a)), _p(self._vec(b)))

    def gSumMag(self, a):
        return lib().orc_gSumMag(C.byref(self.sys), _p(self._vec(a)))

    def normFactor(self, psi, source, Apsi):
        tmp = self._vec()
        return lib().orc_normFactor(C.byref(self.sys), _p(self._vec(psi)), _p(self._vec(source)),
                                    _p(self._vec(Apsi)), _p(tmp))

    def precondition(self, kind, r, transpose=False, d=0):
        """rank-local preconditioner of domain d (kind in DIC / DILU)."""
        D = self.doms[d]
        rD = np.zeros(D.nCells); w = np.zeros(D.nCells)
        r = np.ascontiguousarray(r, dtype=np.float64)
        L = lib()
        if kind == "DIC":
            L.orc_DIC_calcReciprocalD(C.byref(D), _p(rD))
            L.orc_DIC_precondition(C.byref(D), _p(rD), _p(w), _p(r))
        elif kind == "DILU":
            L.orc_DILU_calcReciprocalD(C.byref(D), _p(rD))
            (L.orc_DILU_preconditionT if transpose else L.orc_DILU_precondition)(
                C.byref(D), _p(rD), _p(w), _p(r))
        else:
            raise ValueError(kind)
        return w, rD

    def smooth(self, smoother, psi, source, nSweeps):
        x = self._vec(psi); b = self._vec(source)
        lib().orc_smooth(C.byref(self.sys), SMOOTHERS[smoother], _p(x), _p(b), int(nSweeps))
        return x

    # --- coupled family: fields are (nCells, nc) arrays (Field<Type> images)
    def _fld(self, x, nc):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.n, nc).copy()
        return x

    def c_ATmul(self, psi, transpose=False):
        nc = np.asarray(psi).reshape(self.n, -1).shape[1]
        x = self._fld(psi, nc); y = np.zeros_like(x)
        lib().orc_c_ATmul(C.byref(self.sys), nc, _p(y), _p(x), int(transpose)); return y

    def c_residual(self, psi, source):
        nc = np.asarray(psi).reshape(self.n, -1).shape[1]
        x = self._fld(psi, nc); b = self._fld(source, nc); y = np.zeros_like(x)
        lib().orc_c_residual(C.byref(self.sys), nc, _p(y), _p(x), _p(b)); return y

    def c_precondition(self, kind, r, transpose=False):
        nc = np.asarray(r).reshape(self.n, -1).shape[1]
        rr = self._fld(r, nc); w = np.zeros_like(rr)
        lib().orc_c_precondition(C.byref(self.sys), CPRECONDS[kind], nc, _p(w), _p(rr), int(transpose)); return w

    def c_smooth(self, psi, source, nSweeps):
        nc = np.asarray(psi).reshape(self.n, -1).shape[1]
        x = self._fld(psi, nc); b = self._fld(source, nc)
        lib().orc_c_smooth(C.byref(self.sys), nc, _p(x), _p(b), int(nSweeps)); return x

    def c_solve(self, psi, source, solver="PBiCCCG", preconditioner="DILU", tolerance=1e-6, relTol=0.0,
                maxIter=1000, nSweeps=1):
        nc = np.asarray(psi).reshape(self.n, -1).shape[1]
        o = COpts()
        o.solver, o.precond, o.smoother, o.nc = CSOLVERS[solver], CPRECONDS[preconditioner], 0, nc
        o.maxIter, o.nSweeps = int(maxIter), int(nSweeps)
        tol = np.broadcast_to(np.asarray(tolerance, dtype=np.float64), (nc,))
        rel = np.broadcast_to(np.asarray(relTol, dtype=np.float64), (nc,))
        for c in range(nc):
            o.tolerance[c], o.relTol[c] = float(tol[c]), float(rel[c])
        for c, wgt in enumerate([1, 2, 2, 1, 2, 1] if nc == 6 else [1] * 9):   # the `&&` of symmTensor / vector, tensor
            o.ipw[c] = float(wgt)
        x = self._fld(psi, nc); b = self._fld(source, nc)
        perf = CPerf()
        rc = lib().orc_c_solve(C.byref(self.sys), C.byref(o), _p(x), _p(b), C.byref(perf))
        if rc:
            raise ValueError("coupled solver selection error %d (name not in the reference's table)" % rc)
        return x, dict(initialResidual=np.array(perf.initialResidual[:nc]),
                       finalResidual=np.array(perf.finalResidual[:nc]),
                       normFactor=np.array(perf.normFactor[:nc]), nIterations=perf.nIterations,
                       converged=bool(perf.converged), singular=[bool(v) for v in perf.singular[:nc]])

    def face_weights(self):
        ws = [np.ascontiguousarray(p.get("faceWeights", np.zeros(len(p["lowerAddr"]))), dtype=np.float64)
              for p in self.problems]
        return np.concatenate(ws) if ws else np.zeros(0)

    def solve(self, psi, source, **optkw):
        o = make_opts(**optkw)
        x = self._vec(psi); b = self._vec(source)
        hist = np.zeros(o.maxIter + 3)
        fw = self.face_weights()
        perf = lib().orc_solve(C.byref(self.sys), C.byref(o), _p(x), _p(b), _p(fw), _p(hist))
        return x, dict(initialResidual=perf.initialResidual, finalResidual=perf.finalResidual,
                       normFactor=perf.normFactor, nIterations=perf.nIterations,
                       converged=bool(perf.converged), singular=bool(perf.singular),
                       history=hist[:perf.nHist].copy())

    def time_gamg_vcycles(self, rA, nVcycles=2, **optkw):
        """CPU baseline helper: build the hierarchy once (cacheAgglomeration), then time
        nVcycles V-cycles (GAMGPreconditioner-style application).  Returns (seconds, setup_s)."""
        import time
        o = make_opts(nVcycles=nVcycles, **optkw)
        fw = self.face_weights()
        L = lib()
        L.orc_gamg_pre_new.restype = C.c_void_p
        t0 = time.perf_counter()
        g = C.c_void_p(L.orc_gamg_pre_new(C.byref(self.sys), C.byref(o), _p(fw)))
        t1 = time.perf_counter()
        w = self._vec()
        r = self._vec(rA)
        L.orc_gamg_pre_apply(g, _p(w), _p(r))
        t2 = time.perf_counter()
        L.orc_gamg_pre_free(g)
        return t2 - t1, t1 - t0

    def gamg_level_sizes(self, **optkw):
        """[(cells, faces) summed over the domains] of every coarse level of the K-domain hierarchy"""
        o = make_opts(**optkw)
        fw = self.face_weights()
        L = lib()
        g = C.c_void_p(L.orc_gamg_build(C.byref(self.sys), C.byref(o), _p(fw)))
        out = [(sum(L.orc_gamg_level_nCells_dom(g, lev, d) for d in range(len(self.doms))),
                sum(L.orc_gamg_level_nFaces_dom(g, lev, d) for d in range(len(self.doms))))
               for lev in range(L.orc_gamg_nLevels(g))]
        L.orc_gamg_free(g)
        return out

    def gamg_levels(self, **optkw):
        o = make_opts(**optkw)
        fw = self.face_weights()
        L = lib()
        g = L.orc_gamg_build(C.byref(self.sys), C.byref(o), _p(fw))
        g = C.c_void_p(g)
        out = []
        nFineC, nFineF = self.doms[0].nCells, self.doms[0].nFaces
        sym = self.sym
        for lev in range(L.orc_gamg_nLevels(g)):
            nc, nf = L.orc_gamg_level_nCells(g, lev), L.orc_gamg_level_nFaces(g, lev)

            def arr(fn, n, ct):
                ptr = getattr(L, fn)(g, lev)
                if not ptr or n == 0:
                    return np.zeros(0, dtype=np.int32 if ct is C.c_int else np.float64)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy()
            d = dict(nCells=nc, restrict=arr("orc_gamg_restrict", nFineC, C.c_int),
                     faceRestrict=arr("orc_gamg_faceRestrict", nFineF, C.c_int),
                     lowerAddr=arr("orc_gamg_level_lower", nf, C.c_int),
                     upperAddr=arr("orc_gamg_level_upper", nf, C.c_int),
                     diag=arr("orc_gamg_level_diag", nc, C.c_double),
                     upper=arr("orc_gamg_level_upperCoeffs", nf, C.c_double))
            if not sym:
                d["lower"] = arr("orc_gamg_level_lowerCoeffs", nf, C.c_double)
            out.append(d)
            nFineC, nFineF = nc, nf
        L.orc_gamg_free(g)
        return out


# ---------------------------------------------------------------- the real reference

def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "ref_driver"))


def dict_string(**kw):
    """OpenFOAM dictionary text for a solver-controls dict (fvSolution syntax)."""
    return " ".join("%s %s;" % (k, ("on" if v else "off") if isinstance(v, bool) else v)
                    for k, v in kw.items())


def run_ref(mode, problem, dict_str=""):
    """Run oracle/_ref/ref_driver (the reference's own libOpenFOAM) on a problem.
    Returns (arrays, stdout)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_ldub", os.path.join(HERE, "..", "openfoam-2.2.x_amd", "ldub.py"))
    ldub = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ldub)
    d = tempfile.mkdtemp(prefix="ldu_ref_")
    arrays = {"nCells": np.array([problem["nCells"]], dtype=np.int32)}
    for k in ("lowerAddr", "upperAddr", "diag", "upper", "lower", "source", "psi", "faceWeights", "psiV",
              "sourceV"):
        if k in problem:
            arrays[k] = problem[k]
    ldub.write(os.path.join(d, "p.ldub"), arrays)
    env = dict(os.environ, WM_PROJECT="OpenFOAM", WM_PROJECT_VERSION="2.2.x",
               WM_PROJECT_DIR=os.path.join(HERE, "_ref"))
    r = subprocess.run([os.path.join(HERE, "_ref", "ref_driver"), mode, os.path.join(d, "p.ldub"),
                        os.path.join(d, "o.ldub"), os.path.join(d, "case"), dict_str],
                       env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_driver failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    out = ldub.read(os.path.join(d, "o.ldub"))
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    return out, r.stdout


def parse_history(stdout, solver_name):
    """Per-iteration residuals printed by SolverPerformance::checkConvergence (debug>=2):
    '<name>:  Iteration N residual = r' (SolverPerformance.C:65-71)."""
    pat = re.compile(r"^%s:  Iteration (\d+) residual = (\S+)" % re.escape(solver_name), re.M)
    return np.array([float(m.group(2)) for m in pat.finditer(stdout)])
