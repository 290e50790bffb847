"""Bandwidth of the finite-volume / fvMatrix-glue kernels at the benchmark's size (216^3 box: 10.08 M cells, 30.09 M
internal faces, 6 x 216^2 boundary faces), device-resident fields: wall time per C-ABI call (each call ends with a
stream sync) over `reps` calls -> achieved GB/s against the ALGORITHMIC bytes of the operation (every array read or
written once, f64 = 8 B, i32 = 4 B; gathers through owner/neighbour counted as one read of the field).
python tools/fv_probe.py [n=216] [reps=10]"""
import ctypes as C
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import __graft_entry__ as entry
entry.load_package()
import torch
from openfoam_amd import capi, cases

n = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
l, u, d = cases.box_addressing(n, n, n)
nC, nF = n ** 3, l.size
ctx = capi.Context(0)
a = capi.Addressing(ctx, nC, l, u)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
R = lambda *shape: torch.rand(*shape, dtype=torch.float64, device=dev, generator=g) + 0.5
Z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)
# boundary: the six faces of the box, one patch each
c = np.arange(nC); i, j, k = c % n, (c // n) % n, c // (n * n)
fcs = [c[i == 0], c[i == n - 1], c[j == 0], c[j == n - 1], c[k == 0], c[k == n - 1]]
b = capi.FvBoundary(a, [x.astype(np.int32) for x in fcs])
nB = b.n
L = capi.lib()
P = capi._ptr
vf, vf3, g3, g9, V = R(nC), R(nC, 3), R(nC, 3), R(nC, 9), R(nC)
w, delta, gms, phi, Sf, cv = R(nF), R(nF), R(nF), R(nF) - 1.0, R(nF, 3), R(nF, 3) - 1.0
C3, Cf3 = R(nC, 3), R(nF, 3)
bS, bV, bV3, bSf = R(nB), R(nB), R(nB, 3), R(nB, 3)
iC, bC = R(nB), R(nB)
upper, lower, diag, source = -R(nF), -R(nF), R(nC) * 8, R(nC)
oF, oF3, oC, oC3, oC9 = Z(nF), Z(nF, 3), Z(nC), Z(nC, 3), Z(nC, 9)
own_t, nei_t = torch.from_numpy(l).to(dev), torch.from_numpy(u).to(dev)

OVERHEAD = [0.0]


def wall(fn, name=""):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = fn()
        assert rc == 0, (name, L.ldu_last_error())
    return (time.perf_counter() - t0) / reps


def bench(name, nbytes, fn):
    dt = wall(fn, name)
    k = max(dt - OVERHEAD[0], 1e-9)
    print("%-46s %8.3f ms  %7.1f GB/s  (%5.1f%% of 8 TB/s) | without the call overhead: %7.3f ms  %5.1f%%"
          % (name, dt * 1e3, nbytes / dt / 1e9, 100 * nbytes / dt / 8e12, k * 1e3, 100 * nbytes / k / 8e12), flush=True)

B8 = 8.0
# every C-ABI call ends with a stream synchronisation: launch + sync of the same entry point on a 2-cell mesh is the overhead each
# wall time below contains (rocprofv3's kernel times of this script: profiles/r03_fv_probe_kernel_stats.csv)
_l0, _u0 = np.array([0], dtype=np.int32), np.array([1], dtype=np.int32)
_a0 = capi.Addressing(ctx, 2, _l0, _u0)
_w0, _v0, _o0 = R(1), R(2), Z(1)
OVERHEAD[0] = min(wall(lambda: L.ldu_fv_interpolate(_a0.h, 1, P(_w0), P(_v0), P(_o0))) for _ in range(3))
print("call overhead (launch + stream sync, ldu_fv_interpolate on a 2-cell mesh): %.1f us" % (OVERHEAD[0] * 1e6))
print("fv / glue kernels, %d^3 box: %d cells, %d faces, %d boundary faces, %d reps" % (n, nC, nF, nB, reps))
bench("ldu_fv_interpolate (scalar)", nF * (8 + 8 + B8) + nC * 8, lambda: L.ldu_fv_interpolate(a.h, 1, P(w), P(vf), P(oF)))
bench("ldu_fv_interpolate (vector)", nF * (8 + 8 + 24) + nC * 24, lambda: L.ldu_fv_interpolate(a.h, 3, P(w), P(vf3), P(oF3)))
bench("ldu_fvc_surfaceIntegrate (scalar)", nF * (8 + 8) + nC * 16 + nC * 8, lambda: L.ldu_fvc_surfaceIntegrate(a.h, 1, P(phi), P(V), P(oC)))
bench("ldu_fvc_surfaceIntegrateFull (vector, patches)", nF * (24 + 8) + nC * 32 + nB * 28, lambda: L.ldu_fvc_surfaceIntegrateFull(a.h, b.h, 3, P(oF3), P(bV3), P(V), P(oC3)))
bench("ldu_fvc_gaussGradFull (scalar)", nF * (24 + 8 + 8) + nC * 32 + nB * 36, lambda: L.ldu_fvc_gaussGradFull(a.h, b.h, 1, P(Sf), P(phi), P(bSf), P(bS), P(V), P(oC3)))
bench("ldu_fvc_gaussGradFull (vector)", nF * (24 + 24 + 8) + nC * 80 + nB * 52, lambda: L.ldu_fvc_gaussGradFull(a.h, b.h, 3, P(Sf), P(oF3), P(bSf), P(bV3), P(V), P(oC9)))
bench("ldu_fvc_snGrad", nF * 24 + nC * 8, lambda: L.ldu_fvc_snGrad(a.h, P(delta), P(vf), P(oF)))
bench("ldu_fv_interpolateDot (corrVec & grad, scalar)", nF * (24 + 8 + 8 + 8) + nC * 24, lambda: L.ldu_fv_interpolateDot(a.h, 3, P(cv), P(w), P(g3), P(oF)))
bench("ldu_fv_interpolateDot (Sf & tensor)", nF * (24 + 8 + 8 + 24) + nC * 72, lambda: L.ldu_fv_interpolateDot(a.h, 9, P(Sf), P(w), P(g9), P(oF3)))
bench("ldu_fvc_correctedSnGrad (scalar)", nF * (8 + 8 + 8 + 8) + nC * 8, lambda: L.ldu_fvc_correctedSnGrad(a.h, 1, P(delta), P(vf), P(phi), P(oF)))
bench("ldu_fvm_sourceMinusVDiv (scalar, patches)", nF * 16 + nC * 32 + nB * 12, lambda: L.ldu_fvm_sourceMinusVDiv(a.h, b.h, 1, P(phi), P(bS), P(V), P(oC)))
bench("ldu_fvm_laplacian", nF * (8 + 8 + 8 + 8) + nC * 16, lambda: L.ldu_fvm_laplacian(a.h, P(delta), P(gms), P(oC), P(oF)))
bench("ldu_fvm_div", nF * (8 + 8 + 16 + 8) + nC * 16, lambda: L.ldu_fvm_div(a.h, P(w), P(phi), P(oC), P(oF), P(lower)))
bench("ldu_mesh_nonorth_factors", nF * (8 + 24 + 8 + 8 + 24) + nC * 24, lambda: L.ldu_mesh_nonorth_factors(ctx.h, nC, nF, P(own_t), P(nei_t), P(Sf), P(w), P(C3), P(oF), P(oF3)))
bench("ldu_fv_linearUpwindCorrection", nF * (8 + 8 + 24 + 8) + nC * 48, lambda: L.ldu_fv_linearUpwindCorrection(a.h, P(phi), P(C3), P(Cf3), P(g3), P(oF)))
bench("ldu_fvm_addBoundaryDiag", nB * 12 + 2 * 8 * nB, lambda: L.ldu_fvm_addBoundaryDiag(b.h, P(iC), P(oC)))
bench("ldu_fvm_relax (asym)", nF * (16 + 8) + nC * (8 * 4 + 8) + nB * 28, lambda: L.ldu_fvm_relax(b.h, C.c_double(0.7), P(iC), P(bC), P(upper), P(lower), P(vf), P(diag), P(source)))
bench("ldu_fvm_H (asym)", nF * (16 + 8) + nC * 40 + nB * 28, lambda: L.ldu_fvm_H(b.h, P(iC), P(bC), None, P(upper), P(lower), P(vf), P(source), P(V), P(oC)))
bench("ldu_fvm_flux (asym)", nF * (16 + 8 + 8) + nC * 8 + nB * 36, lambda: L.ldu_fvm_flux(b.h, P(iC), P(bC), None, P(upper), P(lower), P(vf), P(oF), P(bS)))
bench("ldu_fvm_A", nC * 24 + nB * 12, lambda: L.ldu_fvm_A(b.h, P(iC), P(diag), P(V), P(oC)))
