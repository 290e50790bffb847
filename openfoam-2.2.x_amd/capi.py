"""ctypes binding of libldugpu.so (include/ldugpu.h) - the host-side Python mirror used by
the tests and bench.py.  Thin: it only marshals numpy arrays / device pointers (torch tensors'
data_ptr) into the C ABI.  Fails loudly when the HIP library is missing or no GPU is visible:
there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libldugpu.so")

SOLVERS = {"PCG": 0, "PBiCG": 1, "smoothSolver": 2, "GAMG": 3, "diagonal": 4}
PRECONDITIONERS = {"none": 0, "diagonal": 1, "DIC": 2, "FDIC": 3, "DILU": 4, "GAMG": 5}
SMOOTHERS = {"GaussSeidel": 0, "symGaussSeidel": 1, "DIC": 2, "DILU": 3, "DICGaussSeidel": 4,
             "DILUGaussSeidel": 5, "FDIC": 6, "nonBlockingGaussSeidel": 7}
AGGLOMERATORS = {"faceAreaPair": 0, "algebraicPair": 1}

EXPORTS = [
    "ldu_last_error", "ldu_default_controls", "ldu_ctx_create", "ldu_ctx_destroy", "ldu_ctx_sync",
    "ldu_ctx_set_spin_limit", "ldu_ctx_fallback_count", "ldu_ctx_overlapped_halo_count", "ldu_debug_div_check", "ldu_debug_cluster_trace", "ldu_debug_cluster_levels", "ldu_debug_gs_multi_trace", "ldu_debug_blocks_trace", "ldu_debug_blocks_info", "ldu_debug_gs_layouts", "ldu_debug_slice_levels",
    "ldu_comm_unique_id", "ldu_ctx_comm_init", "ldu_ctx_comm_init_local", "ldu_ctx_comm_init_peer", "ldu_ctx_comm_select", "ldu_ctx_comm_info", "ldu_addr_create", "ldu_addr_add_patch",
    "ldu_addr_add_cyclic_patch", "ldu_addr_sweep_engine", "ldu_addr_finalize", "ldu_addr_destroy", "ldu_addr_info", "ldu_addr_set_face_weights", "ldu_addr_set_subdomains", "ldu_partition_blobs", "ldu_partition_blobs_footprint",
    "ldu_addr_set_face_areas", "ldu_addr_get_face_weights", "ldu_device_count",
    "ldu_matrix_create", "ldu_matrix_destroy", "ldu_matrix_set_coeffs", "ldu_matrix_set_patch_coeffs",
    "ldu_amul", "ldu_tmul", "ldu_sumA", "ldu_residual", "ldu_H", "ldu_H1", "ldu_faceH",
    "ldu_gSumProd", "ldu_gSumMag", "ldu_precondition", "ldu_smooth", "ldu_solve", "ldu_gamg_levels",
    "ldu_gamg_level_data", "ldu_gamg_level_info", "ldu_fv_interpolate", "ldu_fvc_surfaceIntegrate", "ldu_fvc_gaussGrad",
    "ldu_fvc_snGrad", "ldu_fvm_laplacian", "ldu_fvm_div", "ldu_profile_begin", "ldu_profile_end",
    "ldu_fv_boundary_create", "ldu_fv_boundary_destroy", "ldu_fvm_addBoundaryDiag", "ldu_fvm_addBoundarySource",
    "ldu_fvm_relax", "ldu_fvm_setReference", "ldu_fvm_A", "ldu_fvm_H", "ldu_fvm_flux",
    "ldu_fvm_addBoundaryDiagCmpt", "ldu_fvm_addBoundarySourceV", "ldu_fvm_relaxV", "ldu_fvm_AV", "ldu_fvm_HV",
    "ldu_fv_linearUpwindCorrection", "ldu_fvc_cellLimitedGrad",
    "ldu_coupled_default_controls", "ldu_coupled_solve", "ldu_coupled_amul", "ldu_coupled_residual",
    "ldu_coupled_precondition", "ldu_coupled_smooth",
    "ldu_fv_linearUpwindVCorrection", "ldu_fvc_cellLimitedGradV", "ldu_fvm_boundedSp", "ldu_fvc_gaussGradFull",
    "ldu_mesh_nonorth_factors", "ldu_mesh_patch_nonorth_factors", "ldu_fv_interpolateDot", "ldu_fv_faceDot",
    "ldu_fv_faceScale", "ldu_fvc_correctedSnGrad", "ldu_fv_interpolateBoundary", "ldu_fvc_gaussGradBoundary",
    "ldu_fvc_surfaceIntegrateFull", "ldu_fvm_sourceMinusVDiv", "ldu_fv_tensorGammaFactors",
    "ldu_mesh_geometry", "ldu_mesh_interpolation_factors", "ldu_band_compression", "ldu_tile_shuffle", "ldu_matrix_wait_plans", "ldu_renumber_addressing",
    "ldu_debug_dag_stats", "ldu_debug_stream", "ldu_debug_slices", "ldu_ctx_set_watchdog", "ldu_ctx_comm_counters", "ldu_comm_paired_patch", "ldu_comm_exchange_order",
]

# LduMatrix<Type, scalar, scalar> run-time selection names (Solvers/*/*.H TypeName)
COUPLED_SOLVERS = {"PCICG": 0, "PBiCCCG": 1, "PBiCICG": 2, "SmoothSolver": 3, "diagonal": 4}
COUPLED_PRECONDITIONERS = {"none": 0, "diagonal": 1, "DILU": 2}
COUPLED_SMOOTHERS = {"GaussSeidel": 0}


class CoupledControls(C.Structure):
    _fields_ = [("solver", C.c_int32), ("preconditioner", C.c_int32), ("smoother", C.c_int32),
                ("nCmpt", C.c_int32), ("maxIter", C.c_int32), ("nSweeps", C.c_int32),
                ("tolerance", C.c_double * 9), ("relTol", C.c_double * 9), ("innerProductWeights", C.c_double * 9)]


class CoupledPerf(C.Structure):
    _fields_ = [("initialResidual", C.c_double * 9), ("finalResidual", C.c_double * 9),
                ("normFactor", C.c_double * 9), ("singular", C.c_int32 * 9),
                ("nIterations", C.c_int32), ("converged", C.c_int32), ("solveSeconds", C.c_double)]


class Controls(C.Structure):
    _fields_ = [("solver", C.c_int32), ("preconditioner", C.c_int32), ("smoother", C.c_int32),
                ("tolerance", C.c_double), ("relTol", C.c_double), ("maxIter", C.c_int32),
                ("nSweeps", C.c_int32), ("cacheAgglomeration", C.c_int32),
                ("nPreSweeps", C.c_int32), ("preSweepsLevelMultiplier", C.c_int32),
                ("maxPreSweeps", C.c_int32), ("nPostSweeps", C.c_int32),
                ("postSweepsLevelMultiplier", C.c_int32), ("maxPostSweeps", C.c_int32),
                ("nFinestSweeps", C.c_int32), ("interpolateCorrection", C.c_int32),
                ("scaleCorrection", C.c_int32), ("directSolveCoarsest", C.c_int32),
                ("nCellsInCoarsestLevel", C.c_int32), ("mergeLevels", C.c_int32),
                ("agglomerator", C.c_int32), ("nVcycles", C.c_int32), ("historyCapacity", C.c_int32)]


class Perf(C.Structure):
    _fields_ = [("initialResidual", C.c_double), ("finalResidual", C.c_double),
                ("normFactor", C.c_double), ("nIterations", C.c_int32), ("converged", C.c_int32),
                ("singular", C.c_int32), ("nHistory", C.c_int32), ("solveSeconds", C.c_double),
                ("setupSeconds", C.c_double)]


_LIB = None


def build_library():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "csrc"), "-j8"])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libldugpu.so is not built (%s): run __graft_entry__.build(); "
                               "there is no CPU fallback for the lduMatrix GPU path" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.ldu_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


class LduError(RuntimeError):
    pass


def _chk(rc):
    if rc != 0:
        raise LduError("libldugpu error %d: %s" % (rc, lib().ldu_last_error().decode()))


def _ptr(x):
    """numpy array, torch tensor, int address or None -> c_void_p"""
    if x is None:
        return C.c_void_p(0)
    if isinstance(x, np.ndarray):
        return C.c_void_p(x.ctypes.data)
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def _f64(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.float64)
    return x


def make_controls(**kw):
    c = Controls()
    lib().ldu_default_controls(C.byref(c))
    for k, v in kw.items():
        if k == "solver":
            v = SOLVERS[v]
        elif k == "preconditioner":
            v = PRECONDITIONERS[v]
        elif k == "smoother":
            v = SMOOTHERS[v]
        elif k == "agglomerator":
            v = AGGLOMERATORS[v]
        if isinstance(v, bool):
            v = int(v)
        setattr(c, k, v)
    return c


OOB_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                     C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


def oob_torch(group=None):
    """out-of-band exchange over a torch.distributed group of CPU tensors (gloo): isend / irecv per peer"""
    import torch
    import torch.distributed as dist

    def exchange(peers, send, recv_sizes):
        bufs = [torch.empty(max(1, n), dtype=torch.uint8) for n in recv_sizes]
        ops = []
        for p, b, n, r in zip(peers, send, recv_sizes, bufs):
            t = torch.frombuffer(bytearray(b), dtype=torch.uint8) if len(b) else torch.zeros(1, dtype=torch.uint8)
            ops.append(dist.P2POp(dist.isend, t, dist.get_global_rank(group, p) if group is not None else p, group))
            ops.append(dist.P2POp(dist.irecv, r, dist.get_global_rank(group, p) if group is not None else p, group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        return [bytes(r[:n].numpy().tobytes()) for r, n in zip(bufs, recv_sizes)]
    return exchange


class OobThreads:
    """out-of-band exchange between the threads of ONE process (ranks as threads on one GPU, tests): a mailbox per
    (source, destination) pair"""

    def __init__(self, n):
        import threading
        self.n = n
        self.cv = threading.Condition()
        self.box = {}

    def for_rank(self, me):
        def exchange(peers, send, recv_sizes):
            with self.cv:
                for p, b in zip(peers, send):
                    self.box.setdefault((me, p), []).append(b)
                self.cv.notify_all()
            out = []
            for p in peers:
                with self.cv:
                    self.cv.wait_for(lambda: self.box.get((p, me)), timeout=120)
                    out.append(self.box[(p, me)].pop(0))
            return out
        return exchange


class Context:
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().ldu_ctx_create(C.byref(self.h), int(device)))

    def comm_init(self, rank, n_ranks, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _chk(lib().ldu_ctx_comm_init(self.h, int(rank), int(n_ranks), buf))

    def comm_init_local(self, rank, n_ranks, group_id):
        """Test facility: ranks = threads of this process sharing one GPU (no RCCL)."""
        _chk(lib().ldu_ctx_comm_init_local(self.h, int(rank), int(n_ranks), int(group_id)))

    def comm_init_peer(self, rank, n_ranks, exchange=None):
        """Peer-store backend (ldu_ctx_comm_init_peer).  exchange(peers, send_blobs, recv_sizes) -> list of bytes: the
        pairwise out-of-band exchange of the set-up messages (oob_torch / OobThreads below); None for one rank."""
        cb = None
        if exchange is not None:
            def _cb(user, n, peers, sbufs, sbytes, rbufs, rbytes):
                try:
                    pl = [int(peers[i]) for i in range(n)]
                    send = [C.string_at(sbufs[i], int(sbytes[i])) for i in range(n)]
                    got = exchange(pl, send, [int(rbytes[i]) for i in range(n)])
                    for i in range(n):
                        if len(got[i]) != int(rbytes[i]):
                            return 2
                        C.memmove(rbufs[i], got[i], len(got[i]))
                    return 0
                except Exception as e:  # pragma: no cover
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = OOB_FN(_cb)
        self._oob_cb = cb   # keep the trampoline alive as long as the context
        f = lib().ldu_ctx_comm_init_peer
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _chk(f(self.h, int(rank), int(n_ranks), C.cast(cb, C.c_void_p) if cb else None, None))

    def comm_select(self, peer_halo, peer_reduce):
        """which operations travel by peer stores when an RCCL communicator exists as well (ldu_ctx_comm_select)"""
        _chk(lib().ldu_ctx_comm_select(self.h, int(bool(peer_halo)), int(bool(peer_reduce))))

    def comm_info(self):
        out = (C.c_int32 * 4)()
        _chk(lib().ldu_ctx_comm_info(self.h, out))
        return dict(rccl_ranks=out[0], peer_ranks=out[1], halo="peer stores" if out[2] else ("rccl" if out[0] else None),
                    sums="peer stores" if out[3] else ("rccl" if out[0] else None))

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _chk(lib().ldu_comm_unique_id(buf))
        return bytes(buf)

    def sync(self):
        _chk(lib().ldu_ctx_sync(self.h))

    def set_spin_limit(self, polls):
        """bound of the sweep engines' dependency waits (0 = default); tiny values force the engine fallback"""
        _chk(lib().ldu_ctx_set_spin_limit(self.h, C.c_uint32(int(polls))))

    def overlapped_halo_count(self):
        f = lib().ldu_ctx_overlapped_halo_count
        f.restype = C.c_int64
        return int(f(self.h))

    def div_check(self, n, seed=1):
        """ldu_debug_div_check: mismatches between the sweep kernels' split division and the compiler's (must be 0)."""
        bad = C.c_uint64(0)
        f = lib().ldu_debug_div_check
        f.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_uint64)]
        _chk(f(self.h, seed, n, C.byref(bad)))
        return int(bad.value)

    def stream(self, mode, n, reps=10):
        """ldu_debug_stream: GB/s of the copy (mode 0, 16 B/element) or triad (mode 1, 24 B/element) kernel on n doubles"""
        sec = C.c_double()
        f = lib().ldu_debug_stream
        f.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p]
        _chk(f(self.h, int(mode), int(n), int(reps), C.byref(sec)))
        return (16.0 if mode == 0 else 24.0) * (int(n) // 2 * 2) / sec.value / 1e9

    def set_watchdog(self, budget_ms=200.0, debug_stall_ms=0.0):
        f = lib().ldu_ctx_set_watchdog
        f.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _chk(f(self.h, float(budget_ms), float(debug_stall_ms)))

    def comm_counters(self):
        out = (C.c_int64 * 4)()
        f = lib().ldu_ctx_comm_counters
        f.argtypes = [C.c_void_p, C.c_void_p]
        _chk(f(self.h, out))
        return dict(halo_exchanges=out[0], all_reduces=out[1], scalar_readbacks=out[2], halo_overlapped=out[3])

    def fallback_count(self):
        f = lib().ldu_ctx_fallback_count
        f.restype = C.c_int64
        return int(f(self.h))

    def close(self):
        if self.h:
            lib().ldu_ctx_destroy(self.h)
            self.h = C.c_void_p()


class Addressing:
    """Device image of one lduAddressing (+ coupled patches)."""

    def __init__(self, ctx, nCells, lowerAddr, upperAddr, faceWeights=None, patches=(), subdomains=None):
        self.ctx = ctx
        self.nCells = int(nCells)
        l = np.ascontiguousarray(lowerAddr, dtype=np.int32)
        u = np.ascontiguousarray(upperAddr, dtype=np.int32)
        self.nFaces = int(l.size)
        self.h = C.c_void_p()
        _chk(lib().ldu_addr_create(ctx.h, C.byref(self.h), self.nCells, self.nFaces, _ptr(l), _ptr(u)))
        for p in patches:
            fc = np.ascontiguousarray(p["faceCells"], dtype=np.int32)
            if p.get("nbrPatch") is not None and p.get("cyclic"):
                _chk(lib().ldu_addr_add_cyclic_patch(self.h, int(fc.size), _ptr(fc), int(p["nbrPatch"])))
            else:
                _chk(lib().ldu_addr_add_patch(self.h, int(fc.size), _ptr(fc), int(p["nbrRank"])))
        if patches:
            _chk(lib().ldu_addr_finalize(self.h))
        if faceWeights is not None:
            w = np.ascontiguousarray(faceWeights, dtype=np.float64)
            _chk(lib().ldu_addr_set_face_weights(self.h, _ptr(w)))
        if subdomains is not None:
            sd = np.ascontiguousarray(subdomains, dtype=np.int32)
            _chk(lib().ldu_addr_set_subdomains(self.h, int(sd.max()) + 1 if sd.size else 0, _ptr(sd)))

    def linearUpwindCorrection(self, phi, C3, Cf3, grad3):
        out = np.zeros(self.nFaces)
        _chk(lib().ldu_fv_linearUpwindCorrection(self.h, _ptr(_f64(phi)), _ptr(_f64(C3)), _ptr(_f64(Cf3)),
                                                 _ptr(_f64(grad3)), _ptr(out)))
        return out

    def linearUpwindVCorrection(self, phi, w, vf3, C3, Cf3, grad9):
        out = np.zeros((self.nFaces, 3))
        _chk(lib().ldu_fv_linearUpwindVCorrection(self.h, _ptr(_f64(phi)), _ptr(_f64(w)), _ptr(_f64(vf3)), _ptr(_f64(C3)),
                                                  _ptr(_f64(Cf3)), _ptr(_f64(grad9)), _ptr(out)))
        return out

    # ---- non-orthogonal correction / gaussDiv (include/ldugpu.h; ldu_fvschemes.hip)
    def interpolateDot(self, vec, weights, field):
        fld = np.ascontiguousarray(field, dtype=np.float64).reshape(self.nCells, -1)
        nc = fld.shape[1]
        out = np.zeros(self.nFaces) if nc == 3 else np.zeros((self.nFaces, 3))
        _chk(lib().ldu_fv_interpolateDot(self.h, nc, _ptr(_f64(vec)), _ptr(_f64(weights)), _ptr(fld), _ptr(out)))
        return out

    def correctedSnGrad(self, nonOrthDelta, vf, corr=None):
        v = np.ascontiguousarray(vf, dtype=np.float64)
        nc = 1 if v.ndim == 1 else v.shape[1]
        out = np.zeros(self.nFaces) if nc == 1 else np.zeros((self.nFaces, nc))
        _chk(lib().ldu_fvc_correctedSnGrad(self.h, nc, _ptr(_f64(nonOrthDelta)), _ptr(v), _ptr(_f64(corr)), _ptr(out)))
        return out

    def surfaceIntegrateFull(self, ssf, V, boundary=None, boundarySsf=None):
        f = np.ascontiguousarray(ssf, dtype=np.float64)
        nc = 1 if f.ndim == 1 else f.shape[1]
        out = np.zeros(self.nCells) if nc == 1 else np.zeros((self.nCells, nc))
        _chk(lib().ldu_fvc_surfaceIntegrateFull(self.h, boundary.h if boundary else None, nc, _ptr(f),
                                                _ptr(_f64(boundarySsf)), _ptr(_f64(V)), _ptr(out)))
        return out

    def sourceMinusVDiv(self, source, ffc, V, boundary=None, boundaryFfc=None):
        f = np.ascontiguousarray(ffc, dtype=np.float64)
        nc = 1 if f.ndim == 1 else f.shape[1]
        s = np.array(source, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_sourceMinusVDiv(self.h, boundary.h if boundary else None, nc, _ptr(f),
                                           _ptr(_f64(boundaryFfc)), _ptr(_f64(V)), _ptr(s)))
        return s

    def set_face_areas(self, Sf):
        """faceAreaPair weights from the internal faces' area vectors, computed on the device
        (faceAreaPairGAMGAgglomeration.C:48-73); returns them"""
        sf = np.ascontiguousarray(Sf, dtype=np.float64).reshape(-1, 3)
        assert sf.shape[0] == self.nFaces
        _chk(lib().ldu_addr_set_face_areas(self.h, _ptr(sf)))
        w = np.zeros(self.nFaces)
        _chk(lib().ldu_addr_get_face_weights(self.h, _ptr(w)))
        return w

    ENGINES = ("chip-wide point-to-point", "XCD slabs", "clusters", "single wavefront", "level kernels", "one workgroup", "blocks")

    def sweep_engine(self, kind):
        rc = lib().ldu_addr_sweep_engine(self.h, int(kind))
        if rc < 0:
            _chk(rc)
        return self.ENGINES[rc]

    def info(self):
        nl, ns, ne = C.c_int32(), C.c_int32(), C.c_int64()
        _chk(lib().ldu_addr_info(self.h, C.byref(nl), C.byref(ns), C.byref(ne)))
        return dict(nLevels=nl.value, nSlices=ns.value, nEntriesPadded=ne.value)

    def close(self):
        if self.h:
            lib().ldu_addr_destroy(self.h)
            self.h = C.c_void_p()

    # ---- fv stencils (numpy in / numpy out)
    def interpolate(self, lambdas, vf):
        vf = _f64(vf)
        nComp = 1 if vf.ndim == 1 else vf.shape[1]
        sf = np.zeros((self.nFaces,) if nComp == 1 else (self.nFaces, nComp))
        _chk(lib().ldu_fv_interpolate(self.h, nComp, _ptr(_f64(lambdas)), _ptr(vf), _ptr(sf)))
        return sf

    def surfaceIntegrate(self, ssf, V):
        ssf = _f64(ssf)
        nComp = 1 if ssf.ndim == 1 else ssf.shape[1]
        out = np.zeros((self.nCells,) if nComp == 1 else (self.nCells, nComp))
        _chk(lib().ldu_fvc_surfaceIntegrate(self.h, nComp, _ptr(ssf), _ptr(_f64(V)), _ptr(out)))
        return out

    def gaussGrad(self, Sf, ssf, V):
        out = np.zeros((self.nCells, 3))
        _chk(lib().ldu_fvc_gaussGrad(self.h, _ptr(_f64(Sf)), _ptr(_f64(ssf)), _ptr(_f64(V)), _ptr(out)))
        return out

    def snGrad(self, deltaCoeffs, vf):
        out = np.zeros(self.nFaces)
        _chk(lib().ldu_fvc_snGrad(self.h, _ptr(_f64(deltaCoeffs)), _ptr(_f64(vf)), _ptr(out)))
        return out

    def fvmLaplacian(self, deltaCoeffs, gammaMagSf):
        diag, upper = np.zeros(self.nCells), np.zeros(self.nFaces)
        _chk(lib().ldu_fvm_laplacian(self.h, _ptr(_f64(deltaCoeffs)), _ptr(_f64(gammaMagSf)),
                                     _ptr(diag), _ptr(upper)))
        return diag, upper

    def fvmDiv(self, weights, phi):
        diag, upper, lower = np.zeros(self.nCells), np.zeros(self.nFaces), np.zeros(self.nFaces)
        _chk(lib().ldu_fvm_div(self.h, _ptr(_f64(weights)), _ptr(_f64(phi)), _ptr(diag), _ptr(upper),
                               _ptr(lower)))
        return diag, upper, lower


class Matrix:
    """Device image of one lduMatrix; mirrors lduMatrix::solver / preconditioner / smoother."""

    def __init__(self, addr):
        self.addr = addr
        self.h = C.c_void_p()
        _chk(lib().ldu_matrix_create(addr.h, C.byref(self.h)))
        self._keep = None

    def set_coeffs(self, diag, upper, lower=None):
        diag, upper, lower = _f64(diag), _f64(upper), _f64(lower)
        self._keep = (diag, upper, lower)
        _chk(lib().ldu_matrix_set_coeffs(self.h, _ptr(diag), _ptr(upper), _ptr(lower)))

    def set_patch_coeffs(self, patchI, bou, intc):
        bou, intc = _f64(bou), _f64(intc)
        _chk(lib().ldu_matrix_set_patch_coeffs(self.h, int(patchI), _ptr(bou), _ptr(intc)))
        self.addr.ctx.sync()

    def _vec(self):
        return np.zeros(self.addr.nCells)

    def Amul(self, psi):
        y = self._vec()
        _chk(lib().ldu_amul(self.h, _ptr(y), _ptr(_f64(psi))))
        return y

    def Tmul(self, psi):
        y = self._vec()
        _chk(lib().ldu_tmul(self.h, _ptr(y), _ptr(_f64(psi))))
        return y

    def sumA(self):
        y = self._vec()
        _chk(lib().ldu_sumA(self.h, _ptr(y)))
        return y

    def residual(self, psi, source):
        y = self._vec()
        _chk(lib().ldu_residual(self.h, _ptr(y), _ptr(_f64(psi)), _ptr(_f64(source))))
        return y

    def H(self, psi):
        y = self._vec()
        _chk(lib().ldu_H(self.h, _ptr(y), _ptr(_f64(psi))))
        return y

    def H1(self):
        y = self._vec()
        _chk(lib().ldu_H1(self.h, _ptr(y)))
        return y

    def faceH(self, psi):
        y = np.zeros(self.addr.nFaces)
        _chk(lib().ldu_faceH(self.h, _ptr(y), _ptr(_f64(psi))))
        return y

    def gSumProd(self, a, b):
        r = C.c_double()
        _chk(lib().ldu_gSumProd(self.h, _ptr(_f64(a)), _ptr(_f64(b)), C.byref(r)))
        return r.value

    def gSumMag(self, a):
        r = C.c_double()
        _chk(lib().ldu_gSumMag(self.h, _ptr(_f64(a)), C.byref(r)))
        return r.value

    def precondition(self, kind, rA, transpose=False):
        w = self._vec()
        _chk(lib().ldu_precondition(self.h, PRECONDITIONERS[kind], _ptr(w), _ptr(_f64(rA)), int(transpose)))
        return w

    def smooth(self, smoother, psi, source, nSweeps):
        x = np.array(psi, dtype=np.float64, copy=True)
        _chk(lib().ldu_smooth(self.h, SMOOTHERS[smoother], _ptr(x), _ptr(_f64(source)), int(nSweeps)))
        return x

    def solve(self, psi, source, history=True, inplace=False, **controls):
        """psi, source: numpy arrays (psi is copied unless inplace=True: then the C-contiguous float64 array is handed
        to the library as it is, like the OpenFOAM shim hands over psi.begin()) or device tensors (psi updated in place)."""
        c = make_controls(**controls)
        cap = c.maxIter + 3
        hist = np.zeros(cap)
        c.historyCapacity = cap if history else 0
        perf = Perf()
        if isinstance(psi, np.ndarray):
            if inplace:
                if psi.dtype != np.float64 or not psi.flags["C_CONTIGUOUS"]:
                    raise ValueError("inplace solve needs a C-contiguous float64 array")
                x = psi
            else:
                x = np.array(psi, dtype=np.float64, copy=True)
        else:
            x = psi
        _chk(lib().ldu_solve(self.h, C.byref(c), _ptr(x), _ptr(_f64(source)), C.byref(perf), _ptr(hist)))
        n = min(perf.nHistory, cap)
        return x, dict(initialResidual=perf.initialResidual, finalResidual=perf.finalResidual,
                       normFactor=perf.normFactor, nIterations=perf.nIterations,
                       converged=bool(perf.converged), singular=bool(perf.singular),
                       history=hist[:n].copy(), solveSeconds=perf.solveSeconds)

    # ---- coupled family LduMatrix<Type, scalar, scalar>: fields are (nCells, nCmpt) arrays
    def _fld(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(self.addr.nCells, -1)
        return x

    def coupled_Amul(self, psi, transpose=False):
        x = self._fld(psi)
        y = np.zeros_like(x)
        _chk(lib().ldu_coupled_amul(self.h, x.shape[1], _ptr(y), _ptr(x), int(transpose)))
        return y

    def coupled_residual(self, psi, source):
        x = self._fld(psi)
        y = np.zeros_like(x)
        _chk(lib().ldu_coupled_residual(self.h, x.shape[1], _ptr(y), _ptr(x), _ptr(self._fld(source))))
        return y

    def coupled_precondition(self, kind, rA, transpose=False):
        r = self._fld(rA)
        w = np.zeros_like(r)
        _chk(lib().ldu_coupled_precondition(self.h, COUPLED_PRECONDITIONERS[kind], r.shape[1], _ptr(w), _ptr(r),
                                            int(transpose)))
        return w

    def coupled_smooth(self, psi, source, nSweeps, smoother="GaussSeidel"):
        x = self._fld(psi).copy()
        _chk(lib().ldu_coupled_smooth(self.h, COUPLED_SMOOTHERS[smoother], x.shape[1], _ptr(x),
                                      _ptr(self._fld(source)), int(nSweeps)))
        return x

    def coupled_solve(self, psi, source, solver="PBiCCCG", preconditioner="DILU", smoother="GaussSeidel",
                      tolerance=1e-6, relTol=0.0, maxIter=1000, nSweeps=1):
        """LduMatrix<Type,scalar,scalar>::solver::New(...)->solve(psi) ("type coupled;"); tolerance / relTol are
        Type-valued in the reference's dictionary: scalars are broadcast to every component."""
        x = self._fld(psi).copy()
        nc = x.shape[1]
        c = CoupledControls()
        lib().ldu_coupled_default_controls(C.byref(c), nc)
        c.solver, c.preconditioner = COUPLED_SOLVERS[solver], COUPLED_PRECONDITIONERS[preconditioner]
        c.smoother = COUPLED_SMOOTHERS[smoother]
        c.maxIter, c.nSweeps = int(maxIter), int(nSweeps)
        tol = np.broadcast_to(np.asarray(tolerance, dtype=np.float64), (nc,))
        rel = np.broadcast_to(np.asarray(relTol, dtype=np.float64), (nc,))
        for i in range(nc):
            c.tolerance[i], c.relTol[i] = float(tol[i]), float(rel[i])
        perf = CoupledPerf()
        _chk(lib().ldu_coupled_solve(self.h, C.byref(c), _ptr(x), _ptr(self._fld(source)), C.byref(perf)))
        return x, dict(initialResidual=np.array(perf.initialResidual[:nc]),
                       finalResidual=np.array(perf.finalResidual[:nc]),
                       normFactor=np.array(perf.normFactor[:nc]), nIterations=perf.nIterations,
                       converged=bool(perf.converged), singular=[bool(v) for v in perf.singular[:nc]],
                       solveSeconds=perf.solveSeconds)

    def profile_begin(self):
        _chk(lib().ldu_profile_begin(self.h))

    def profile_end(self):
        ms = (C.c_double * 8)()
        cnt = (C.c_int64 * 8)()
        _chk(lib().ldu_profile_end(self.h, ms, cnt))
        names = ["amul", "gs_sweep", "tri_sweep", "residual", "gs_multi", "c5", "c6", "rd_sweep"]
        return {n: dict(ms=ms[i], count=cnt[i]) for i, n in enumerate(names) if cnt[i]}

    def gs_layouts(self):
        """per-sweep layouts of the chip-wide pipelined GaussSeidel sweeps: dict(built=sweeps with their own layout, slices=[...])"""
        o = np.zeros(6, dtype=np.int64)
        lib().ldu_debug_gs_layouts.argtypes = [C.c_void_p, C.c_void_p]
        _chk(lib().ldu_debug_gs_layouts(self.h, o.ctypes.data))
        return dict(built=int(o[0]), slices=[int(v) for v in o[2:5]], level_slices=int(o[5]))

    def wait_plans(self):
        """sweep plans still being built behind the solves (large GAMG levels): wait for them"""
        _chk(lib().ldu_matrix_wait_plans(self.h))

    def gamg_level_sizes(self, **controls):
        """[(nCells, nFaces, dependency levels, widest row, engine of one GS sweep, engine of pipelined sweeps)] of the
        coarse levels (builds the hierarchy when needed)"""
        c = make_controls(**controls)
        nl = C.c_int32()
        nc = (C.c_int32 * 50)()
        nf = (C.c_int32 * 50)()
        _chk(lib().ldu_gamg_levels(self.h, C.byref(c), C.byref(nl), nc, nf))
        out = []
        for i in range(nl.value):
            info = (C.c_int32 * 8)()
            _chk(lib().ldu_gamg_level_info(self.h, i, info))
            out.append(dict(nCells=info[0], nFaces=info[1], nLevels=info[2], maxRowWidth=info[3],
                            engine_gs=Addressing.ENGINES[info[5]], engine_gs_multi=Addressing.ENGINES[info[6]]))
        return out

    def gamg_levels(self, **controls):
        c = make_controls(**controls)
        nl = C.c_int32()
        nc = (C.c_int32 * 50)()
        nf = (C.c_int32 * 50)()
        _chk(lib().ldu_gamg_levels(self.h, C.byref(c), C.byref(nl), nc, nf))
        out = []
        nFine = self.addr.nCells
        for i in range(nl.value):
            r = np.zeros(nFine, dtype=np.int32)
            d = np.zeros(nc[i])
            up = np.zeros(nf[i])
            lo = np.zeros(nf[i])
            _chk(lib().ldu_gamg_level_data(self.h, i, _ptr(r), _ptr(d), _ptr(up), _ptr(lo)))
            out.append(dict(nCells=nc[i], nFaces=nf[i], restrict=r, diag=d, upper=up, lower=lo))
            nFine = nc[i]
        return out

    def close(self):
        if self.h:
            lib().ldu_matrix_destroy(self.h)
            self.h = C.c_void_p()


def from_problem(ctx, p):
    """cases.py problem dict -> (Addressing, Matrix) with coefficients set."""
    a = Addressing(ctx, p["nCells"], p["lowerAddr"], p["upperAddr"], p.get("faceWeights"),
                   patches=p.get("patches_dev", ()), subdomains=p.get("subdomains"))
    m = Matrix(a)
    m.set_coeffs(p["diag"], p["upper"], p.get("lower"))
    if p.get("patches_dev") and p.get("patches") and "bouCoeffs" in p["patches"][0]:
        for i, q in enumerate(p["patches"]):
            m.set_patch_coeffs(i, q["bouCoeffs"], q["intCoeffs"])
    return a, m


def mesh_nonorth_factors(ctx, nCells, owner, neighbour, Sf, magSf, C):
    """nonOrthDeltaCoeffs, nonOrthCorrectionVectors of the internal faces (surfaceInterpolation.C:289-352)"""
    o = np.ascontiguousarray(owner, dtype=np.int32)
    n = np.ascontiguousarray(neighbour, dtype=np.int32)
    nF = o.size
    nod, cv = np.zeros(nF), np.zeros((nF, 3))
    _chk(lib().ldu_mesh_nonorth_factors(ctx.h, int(nCells), nF, _ptr(o), _ptr(n), _ptr(_f64(Sf)), _ptr(_f64(magSf)),
                                        _ptr(_f64(C)), _ptr(nod), _ptr(cv)))
    return nod, cv


def mesh_patch_nonorth_factors(ctx, Sf, magSf, delta, coupled):
    n = np.asarray(Sf).reshape(-1, 3).shape[0]
    nod, cv = np.zeros(n), np.zeros((n, 3))
    _chk(lib().ldu_mesh_patch_nonorth_factors(ctx.h, n, _ptr(_f64(Sf)), _ptr(_f64(magSf)), _ptr(_f64(delta)),
                                              int(bool(coupled)), _ptr(nod), _ptr(cv)))
    return nod, cv


def fv_face_dot(ctx, vec, field):
    f = np.ascontiguousarray(field, dtype=np.float64)
    n, nc = f.shape
    out = np.zeros(n) if nc == 3 else np.zeros((n, 3))
    _chk(lib().ldu_fv_faceDot(ctx.h, n, nc, _ptr(_f64(vec)), _ptr(f), _ptr(out)))
    return out


def fv_face_scale(ctx, scale, field, accumulate_into=None):
    f = np.ascontiguousarray(field, dtype=np.float64)
    n = f.shape[0]
    nc = 1 if f.ndim == 1 else f.shape[1]
    out = np.zeros_like(f) if accumulate_into is None else np.array(accumulate_into, dtype=np.float64, copy=True)
    _chk(lib().ldu_fv_faceScale(ctx.h, n, nc, _ptr(_f64(scale)), _ptr(f), int(accumulate_into is not None), _ptr(out)))
    return out


def fv_tensor_gamma_factors(ctx, Sf, magSf, gamma):
    g = np.ascontiguousarray(gamma, dtype=np.float64)
    n, ng = g.shape
    sn, sc = np.zeros(n), np.zeros((n, 3))
    _chk(lib().ldu_fv_tensorGammaFactors(ctx.h, n, ng, _ptr(_f64(Sf)), _ptr(_f64(magSf)), _ptr(g), _ptr(sn), _ptr(sc)))
    return sn, sc


class FvBoundary:
    """ldu_fv_boundary: the patches of an fvMesh (sizes, faceCells, coupled flags) for the fvMatrix glue
    (fvMatrix.C addBoundaryDiag/addBoundarySource/relax/A/H/flux).  Coefficient arrays are concatenated
    over the patches in patch order."""

    def __init__(self, addr, face_cells, coupled=None):
        self.addr = addr
        self.sizes = np.array([len(fc) for fc in face_cells], dtype=np.int32)
        fc = (np.concatenate([np.asarray(x, dtype=np.int32) for x in face_cells]) if len(face_cells)
              else np.zeros(0, dtype=np.int32))
        self.fc = np.ascontiguousarray(fc, dtype=np.int32)
        self.n = int(self.fc.size)
        cp = np.zeros(len(face_cells), dtype=np.int32) if coupled is None else np.asarray(coupled, dtype=np.int32)
        self.h = C.c_void_p()
        _chk(lib().ldu_fv_boundary_create(addr.h, len(face_cells), _ptr(self.sizes), _ptr(self.fc), _ptr(cp),
                                          C.byref(self.h)))

    def addBoundaryDiag(self, iC, diag):
        d = np.array(diag, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_addBoundaryDiag(self.h, _ptr(_f64(iC)), _ptr(d)))
        return d

    def interpolateBoundary(self, patchWeights, vf, pnf, values):
        """surfaceInterpolationScheme::interpolate on the patch faces (coupled: w*pif + (1-w)*pnf)"""
        v = np.ascontiguousarray(vf, dtype=np.float64).reshape(self.addr.nCells, -1)
        nc = v.shape[1]
        out = np.zeros((self.n, nc))
        _chk(lib().ldu_fv_interpolateBoundary(self.h, nc, _ptr(_f64(patchWeights)), _ptr(v),
                                              _ptr(np.ascontiguousarray(pnf, dtype=np.float64)),
                                              _ptr(np.ascontiguousarray(values, dtype=np.float64)), _ptr(out)))
        return out[:, 0] if np.asarray(vf).ndim == 1 else out

    def gaussGradBoundary(self, patchNf, grad, patchSnGrad, boundaryGrad):
        """gaussGrad::correctBoundaryConditions on the ordinary patches; coupled faces keep boundaryGrad's values"""
        g = np.ascontiguousarray(grad, dtype=np.float64).reshape(self.addr.nCells, -1)
        nc = g.shape[1] // 3
        out = np.array(boundaryGrad, dtype=np.float64, copy=True).reshape(self.n, 3 * nc)
        _chk(lib().ldu_fvc_gaussGradBoundary(self.h, nc, _ptr(_f64(patchNf)), _ptr(g),
                                             _ptr(np.ascontiguousarray(patchSnGrad, dtype=np.float64)), _ptr(out)))
        return out

    def addBoundarySource(self, bC, pnf, source, couples=True):
        s = np.array(source, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_addBoundarySource(self.h, _ptr(_f64(bC)), _ptr(_f64(pnf)) if pnf is not None else None,
                                             int(couples), _ptr(s)))
        return s

    def relax(self, alpha, iC, bC, upper, lower, psi, diag, source):
        d = np.array(diag, dtype=np.float64, copy=True)
        s = np.array(source, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_relax(self.h, C.c_double(alpha), _ptr(_f64(iC)), _ptr(_f64(bC)), _ptr(_f64(upper)),
                                 _ptr(_f64(lower)) if lower is not None else None, _ptr(_f64(psi)), _ptr(d), _ptr(s)))
        return d, s

    def setReference(self, celli, value, diag, source):
        d = np.array(diag, dtype=np.float64, copy=True)
        s = np.array(source, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_setReference(self.addr.h, int(celli), C.c_double(value), _ptr(d), _ptr(s)))
        return d, s

    def A(self, iC, diag, V):
        out = np.zeros(self.addr.nCells)
        _chk(lib().ldu_fvm_A(self.h, _ptr(_f64(iC)), _ptr(_f64(diag)), _ptr(_f64(V)), _ptr(out)))
        return out

    def H(self, iC, bC, pnf, upper, lower, psi, source, V):
        out = np.zeros(self.addr.nCells)
        _chk(lib().ldu_fvm_H(self.h, _ptr(_f64(iC)), _ptr(_f64(bC)), _ptr(_f64(pnf)) if pnf is not None else None,
                             _ptr(_f64(upper)), _ptr(_f64(lower)) if lower is not None else None, _ptr(_f64(psi)),
                             _ptr(_f64(source)), _ptr(_f64(V)), _ptr(out)))
        return out

    def flux(self, iC, bC, pnf, upper, lower, psi):
        fi = np.zeros(self.addr.nFaces)
        fb = np.zeros(self.n)
        _chk(lib().ldu_fvm_flux(self.h, _ptr(_f64(iC)), _ptr(_f64(bC)), _ptr(_f64(pnf)) if pnf is not None else None,
                                _ptr(_f64(upper)), _ptr(_f64(lower)) if lower is not None else None, _ptr(_f64(psi)),
                                _ptr(fi), _ptr(fb)))
        return fi, fb

    def cellLimitedGrad(self, k, vsf, bVal, C3, Cf3, bCf3, grad3):
        g = np.array(grad3, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvc_cellLimitedGrad(self.addr.h, self.h, C.c_double(k), _ptr(_f64(vsf)), _ptr(_f64(bVal)),
                                           _ptr(_f64(C3)), _ptr(_f64(Cf3)), _ptr(_f64(bCf3)), _ptr(g)))
        return g

    def gaussGradFull(self, Sf3, ssf, bSf3, bssf, V):
        ssf = np.ascontiguousarray(ssf, dtype=np.float64)
        nComp = 1 if ssf.ndim == 1 else ssf.shape[1]
        out = np.zeros((self.addr.nCells, 3 * nComp))
        _chk(lib().ldu_fvc_gaussGradFull(self.addr.h, self.h, nComp, _ptr(_f64(Sf3)), _ptr(ssf), _ptr(_f64(bSf3)),
                                         _ptr(_f64(bssf)), _ptr(_f64(V)), _ptr(out)))
        return out

    def boundedSp(self, phi, bPhi, V, diag):
        d = np.array(diag, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_boundedSp(self.addr.h, self.h, _ptr(_f64(phi)), _ptr(_f64(bPhi)), _ptr(_f64(V)), _ptr(d)))
        return d

    def cellLimitedGradV(self, k, vsf3, bVal3, C3, Cf3, bCf3, grad9):
        g = np.array(grad9, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvc_cellLimitedGradV(self.addr.h, self.h, C.c_double(k), _ptr(_f64(vsf3)), _ptr(_f64(bVal3)),
                                            _ptr(_f64(C3)), _ptr(_f64(Cf3)), _ptr(_f64(bCf3)), _ptr(g)))
        return g

    # ---- vector matrices: coefficient arrays [n][3]
    def addBoundaryDiagCmpt(self, iC3, cmpt, diag):
        d = np.array(diag, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_addBoundaryDiagCmpt(self.h, _ptr(_f64(iC3)), int(cmpt), _ptr(d)))
        return d

    def addBoundarySourceV(self, bC3, pnf3, source3, couples=True):
        s = np.array(source3, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_addBoundarySourceV(self.h, _ptr(_f64(bC3)), _ptr(_f64(pnf3)) if pnf3 is not None else None,
                                              int(couples), _ptr(s)))
        return s

    def relaxV(self, alpha, iC3, bC3, upper, lower, psi3, diag, source3):
        d = np.array(diag, dtype=np.float64, copy=True)
        s = np.array(source3, dtype=np.float64, copy=True)
        _chk(lib().ldu_fvm_relaxV(self.h, C.c_double(alpha), _ptr(_f64(iC3)), _ptr(_f64(bC3)), _ptr(_f64(upper)),
                                  _ptr(_f64(lower)) if lower is not None else None, _ptr(_f64(psi3)), _ptr(d), _ptr(s)))
        return d, s

    def AV(self, iC3, diag, V):
        out = np.zeros(self.addr.nCells)
        _chk(lib().ldu_fvm_AV(self.h, _ptr(_f64(iC3)), _ptr(_f64(diag)), _ptr(_f64(V)), _ptr(out)))
        return out

    def HV(self, iC3, bC3, pnf3, upper, lower, psi3, source3, V):
        out = np.zeros((self.addr.nCells, 3))
        _chk(lib().ldu_fvm_HV(self.h, _ptr(_f64(iC3)), _ptr(_f64(bC3)), _ptr(_f64(pnf3)) if pnf3 is not None else None,
                              _ptr(_f64(upper)), _ptr(_f64(lower)) if lower is not None else None, _ptr(_f64(psi3)),
                              _ptr(_f64(source3)), _ptr(_f64(V)), _ptr(out)))
        return out

    def close(self):
        if self.h:
            lib().ldu_fv_boundary_destroy(self.h)
            self.h = C.c_void_p()


# ---------------------------------------------------------------- mesh side

def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def faces_csr(faces):
    """list of per-face point-label lists -> (faceStart, facePoints)"""
    n = np.array([len(f) for f in faces], dtype=np.int64)
    start = np.zeros(len(faces) + 1, dtype=np.int32)
    start[1:] = np.cumsum(n)
    pts = np.fromiter((p for f in faces for p in f), dtype=np.int32, count=int(start[-1]))
    return start, pts


def mesh_geometry(ctx, points, faceStart, facePoints, owner, neighbour, nCells):
    """primitiveMesh face centres / area vectors (all faces) and cell centres / volumes on the device."""
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    faceStart, facePoints, owner, neighbour = _i32(faceStart), _i32(facePoints), _i32(owner), _i32(neighbour)
    nF = owner.size
    Cf, Sf = np.zeros((nF, 3)), np.zeros((nF, 3))
    C, V = np.zeros((int(nCells), 3)), np.zeros(int(nCells))
    _chk(lib().ldu_mesh_geometry(ctx.h, points.shape[0], _ptr(points), nF, _ptr(faceStart), _ptr(facePoints),
                                 int(nCells), neighbour.size, _ptr(owner), _ptr(neighbour), _ptr(Cf), _ptr(Sf),
                                 _ptr(C), _ptr(V)))
    return Cf, Sf, C, V


def mesh_interpolation_factors(ctx, owner, neighbour, Cf, Sf, C):
    """linear weights, deltaCoeffs and magSf of the internal faces"""
    neighbour = _i32(neighbour)
    nI = neighbour.size
    owner = _i32(np.asarray(owner)[:nI])
    Cf = np.ascontiguousarray(np.asarray(Cf, dtype=np.float64)[:nI])
    Sf = np.ascontiguousarray(np.asarray(Sf, dtype=np.float64)[:nI])
    C = np.ascontiguousarray(C, dtype=np.float64)
    w, d, m = np.zeros(nI), np.zeros(nI), np.zeros(nI)
    _chk(lib().ldu_mesh_interpolation_factors(ctx.h, C.shape[0], nI, _ptr(owner), _ptr(neighbour), _ptr(Cf), _ptr(Sf),
                                              _ptr(C), _ptr(w), _ptr(d), _ptr(m)))
    return w, d, m


def band_compression(nCells, lowerAddr, upperAddr):
    """Foam::bandCompression order (host code in the library): newOrder[i] = old label of new cell i"""
    l, u = _i32(lowerAddr), _i32(upperAddr)
    out = np.zeros(int(nCells), dtype=np.int32)
    _chk(lib().ldu_band_compression(int(nCells), l.size, _ptr(l), _ptr(u), _ptr(out)))
    return out


def tile_shuffle(order, tileSize, seed=1):
    """ldu_tile_shuffle (host code): tiles of tileSize consecutive cells of `order` in hashed order - a manualRenumber list"""
    o = _i32(order)
    out = np.zeros_like(o)
    f = lib().ldu_tile_shuffle
    f.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_uint64, C.c_void_p]
    _chk(f(int(o.size), _ptr(o), int(tileSize), int(seed), _ptr(out)))
    return out


def partition_blobs(nCells, lowerAddr, upperAddr, nParts):
    """sub-domain of every cell: nParts compact breadth-first blobs of nearly equal size (host code in the library)"""
    l, u = _i32(lowerAddr), _i32(upperAddr)
    out = np.zeros(int(nCells), dtype=np.int32)
    _chk(lib().ldu_partition_blobs(int(nCells), l.size, _ptr(l), _ptr(u), int(nParts), _ptr(out)))
    return out


def partition_blobs_footprint(nCells, lowerAddr, upperAddr, slotTarget, maxParts):
    """-> (part of every cell, number of parts): breadth-first blobs of equal footprint (cells + distinct outside neighbours <=
    slotTarget while a blob grows), the block engine's cut of levels whose equal-size blobs do not fit (host code)"""
    l, u = _i32(lowerAddr), _i32(upperAddr)
    out = np.zeros(int(nCells), dtype=np.int32)
    n = C.c_int32(0)
    f = lib().ldu_partition_blobs_footprint
    f.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]
    _chk(f(int(nCells), l.size, _ptr(l), _ptr(u), int(slotTarget), int(maxParts), _ptr(out), C.byref(n)))
    return out, int(n.value)


def renumber_addressing(nCells, lowerAddr, upperAddr, newOrder):
    """-> (newLower, newUpper, faceMap, flip): the addressing after renumbering, upper-triangular order"""
    l, u, o = _i32(lowerAddr), _i32(upperAddr), _i32(newOrder)
    nl, nu, fm = np.zeros_like(l), np.zeros_like(l), np.zeros_like(l)
    fl = np.zeros(l.size, dtype=np.uint8)
    _chk(lib().ldu_renumber_addressing(int(nCells), l.size, _ptr(l), _ptr(u), _ptr(o), _ptr(nl), _ptr(nu), _ptr(fm),
                                       _ptr(fl)))
    return nl, nu, fm, fl


def dag_stats(nCells, lowerAddr, upperAddr, maxCells=64, arrays=False):
    """ldu_debug_dag_stats (host only): dependency levels and cluster partition of an addressing"""
    l, u = _i32(lowerAddr), _i32(upperAddr)
    out = np.zeros(16, dtype=np.int64)
    lev = np.zeros(int(nCells), dtype=np.int32) if arrays else None
    clu = np.zeros(int(nCells), dtype=np.int32) if arrays else None
    cl = np.zeros(int(nCells), dtype=np.int32) if arrays else None
    f = lib().ldu_debug_dag_stats
    f.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _chk(f(int(nCells), l.size, _ptr(l), _ptr(u), int(maxCells), _ptr(out), _ptr(lev) if arrays else None,
           _ptr(clu) if arrays else None, _ptr(cl) if arrays else None))
    keys = ("levels", "clusters", "clusterLevels", "sumDepth", "maxDepth", "maxLower", "maxUpper", "rowsOver6",
            "rowsOver12", "interClusterFaces", "widestLevel", "widestClusterLevel")
    d = {k: int(out[i]) for i, k in enumerate(keys)}
    if arrays:
        d.update(cellLevel=lev, cellCluster=clu, clusterLevel=cl[:d["clusters"]])
    return d


def comm_paired_patch(mine_nbr_ranks, p, their_nbr_ranks, me):
    """ldu_comm_paired_patch (host only): the neighbour's patch that pairs with my patch p (-1: none)"""
    a, b = _i32(mine_nbr_ranks), _i32(their_nbr_ranks)
    f = lib().ldu_comm_paired_patch
    f.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
    return int(f(a.size, _ptr(a), int(p), b.size, _ptr(b), int(me)))


def comm_exchange_order(n_faces, nbr_patch):
    """ldu_comm_exchange_order (host only): patch indices in the order one operator application sends / receives"""
    n, q = _i32(n_faces), _i32(nbr_patch)
    out = np.zeros(n.size, dtype=np.int32)
    f = lib().ldu_comm_exchange_order
    f.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    k = int(f(n.size, _ptr(n), _ptr(q), _ptr(out)))
    if k < 0:
        raise LduError("ldu_comm_exchange_order failed")
    return out[:k]
