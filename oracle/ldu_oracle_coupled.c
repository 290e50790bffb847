/*
 * TEST INFRASTRUCTURE - CPU oracle for the templated coupled solvers
 * LduMatrix<Type, scalar, scalar> (OpenFOAM-2.2.x src/OpenFOAM/matrices/LduMatrix).
 *
 * Plain-C restatement, loop order preserved, -ffp-contract=off.  A Field<Type> is nCells x nc doubles,
 * components interleaved (the memory image of Field<vector>, Field<symmTensor>, ...); for DType = LUType =
 * scalar every dot()/cmptMultiply of the reference is one scalar product per component.
 * Pinned against the real reference by tests/test_oracle_vs_ref.py (ref_driver "coupled*" modes) and the golden
 * vectors tests/golden/coupled_*.npz.  Only tests/ may use it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "ldu_oracle.h"

#define MAXC 9
static const double great_ = 1e20, small_ = 1e-20, vsmall_ = 1e-300;   /* SolverPerformance.H */

enum { ORC_C_PCICG = 0, ORC_C_PBICCCG = 1, ORC_C_PBICICG = 2, ORC_C_SMOOTH = 3, ORC_C_DIAGONAL = 4 };
enum { ORC_CPRE_NONE = 0, ORC_CPRE_DIAGONAL = 1, ORC_CPRE_DILU = 2 };

typedef struct orc_copts {
    int solver, precond, smoother, nc, maxIter, nSweeps;
    double tolerance[MAXC], relTol[MAXC];
    double ipw[MAXC];   /* weights of the Type's `&&`: SymmTensorI.H:212-220 (1 2 2 1 2 1), SphericalTensorI.H:130-133 (3) */
} orc_copts;

typedef struct orc_cperf {
    double initialResidual[MAXC], finalResidual[MAXC], normFactor[MAXC];
    int singular[MAXC];
    int nIterations, converged;
} orc_cperf;

static double stabilise(double x, double y) { return x < 0 ? x - y : x + y; }   /* doubleScalar.H */

/* LduInterfaceField::updateInterfaceMatrix as implemented by the processor / cyclic patch fields:
 * result[faceCells[i]] -= coeffs[i]*pnf[i], pnf = the neighbour's patchInternalField */
static void c_update_interfaces(const orc_sys* s, int d, int nc, double* result, const double* psiAll, int useInt,
                                double sign)
{
    const orc_dom* D = &s->dom[d];
    for (int p = 0; p < D->nPatches; p++)
    {
        const orc_patch* P = &D->patches[p];
        const orc_dom* N = &s->dom[P->nbrDom];
        const orc_patch* NP = &N->patches[P->nbrPatch];
        const double* coeffs = useInt ? P->intCoeffs : P->bouCoeffs;
        const double* psiN = psiAll + (size_t)N->cellOffset * nc;
        for (int i = 0; i < P->n; i++)
        {
            const double cf = sign < 0 ? -coeffs[i] : coeffs[i];
            for (int c = 0; c < nc; c++)
                result[(size_t)P->faceCells[i] * nc + c] -= cf * psiN[(size_t)NP->faceCells[i] * nc + c];
        }
    }
}

/* LduMatrixATmul.C:66-114 (Amul), :117-165 (Tmul) */
void orc_c_ATmul(const orc_sys* s, int nc, double* ApsiAll, const double* psiAll, int transpose)
{
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* Apsi = ApsiAll + (size_t)D->cellOffset * nc;
        const double* psi = psiAll + (size_t)D->cellOffset * nc;
        const double* lo = transpose ? D->upper : D->lower;
        const double* up = transpose ? D->lower : D->upper;
        for (int cell = 0; cell < D->nCells; cell++)
            for (int c = 0; c < nc; c++) Apsi[cell * nc + c] = D->diag[cell] * psi[cell * nc + c];
        for (int face = 0; face < D->nFaces; face++)
            for (int c = 0; c < nc; c++)
            {
                Apsi[D->u[face] * nc + c] += lo[face] * psi[D->l[face] * nc + c];
                Apsi[D->l[face] * nc + c] += up[face] * psi[D->u[face] * nc + c];
            }
        c_update_interfaces(s, d, nc, Apsi, psiAll, transpose, 1.0);
    }
}

/* LduMatrixATmul.C:218-276 */
void orc_c_residual(const orc_sys* s, int nc, double* rAAll, const double* psiAll, const double* sourceAll)
{
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* rA = rAAll + (size_t)D->cellOffset * nc;
        const double* psi = psiAll + (size_t)D->cellOffset * nc;
        const double* b = sourceAll + (size_t)D->cellOffset * nc;
        for (int cell = 0; cell < D->nCells; cell++)
            for (int c = 0; c < nc; c++) rA[cell * nc + c] = b[cell * nc + c] - D->diag[cell] * psi[cell * nc + c];
        for (int face = 0; face < D->nFaces; face++)
            for (int c = 0; c < nc; c++)
            {
                rA[D->u[face] * nc + c] -= D->lower[face] * psi[D->l[face] * nc + c];
                rA[D->l[face] * nc + c] -= D->upper[face] * psi[D->u[face] * nc + c];
            }
        c_update_interfaces(s, d, nc, rA, psiAll, 0, -1.0);
    }
}

/* LduMatrixATmul.C:168-215: every component of sumA is the scalar sumA */
static void c_sumA(const orc_sys* s, double* sumAAll /* scalar per cell */)
{
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* sumA = sumAAll + D->cellOffset;
        for (int cell = 0; cell < D->nCells; cell++) sumA[cell] = D->diag[cell] * 1.0;
        for (int face = 0; face < D->nFaces; face++)
        {
            sumA[D->u[face]] += D->lower[face] * 1.0;
            sumA[D->l[face]] += D->upper[face] * 1.0;
        }
        for (int p = 0; p < D->nPatches; p++)
        {
            const orc_patch* P = &D->patches[p];
            for (int i = 0; i < P->n; i++) sumA[P->faceCells[i]] -= P->bouCoeffs[i] * 1.0;
        }
    }
}

/* per-component reductions: FieldFunctions sumCmptProd / sumCmptMag / sum accumulate cell by cell, ranks
 * are then summed in rank order */
static void c_sumCmptProd(const orc_sys* s, int nc, const double* a, const double* b, double* out)
{
    for (int c = 0; c < nc; c++) out[c] = 0.0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double loc[MAXC] = {0};
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
            for (int c = 0; c < nc; c++) loc[c] += a[(size_t)i * nc + c] * b[(size_t)i * nc + c];
        for (int c = 0; c < nc; c++) out[c] += loc[c];
    }
}
static void c_sumCmptMag(const orc_sys* s, int nc, const double* a, double* out)
{
    for (int c = 0; c < nc; c++) out[c] = 0.0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double loc[MAXC] = {0};
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
            for (int c = 0; c < nc; c++) loc[c] += fabs(a[(size_t)i * nc + c]);
        for (int c = 0; c < nc; c++) out[c] += loc[c];
    }
}
/* gSumProd of two Field<Type>: sum over cells of the double inner product (FieldFunctions.C sumProd, `&&`) */
static double c_sumProd(const orc_sys* s, int nc, const double* ipw, const double* a, const double* b)
{
    double out = 0.0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double loc = 0.0;
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
        {
            double dot = (ipw[0] * a[(size_t)i * nc]) * b[(size_t)i * nc];
            for (int c = 1; c < nc; c++) dot += (ipw[c] * a[(size_t)i * nc + c]) * b[(size_t)i * nc + c];
            loc += dot;
        }
        out += loc;
    }
    return out;
}

/* LduMatrixSolver.C:167-186 */
static void c_normFactor(const orc_sys* s, int nc, const double* psi, const double* source, const double* Apsi,
                         double* nf)
{
    const int n = s->nCellsTotal;
    double* sumA = (double*)malloc(sizeof(double) * (n + 1));
    c_sumA(s, sumA);
    /* gAverage(psi) = gSum/gCount */
    double avg[MAXC];
    for (int c = 0; c < nc; c++) avg[c] = 0.0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double loc[MAXC] = {0};
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
            for (int c = 0; c < nc; c++) loc[c] += psi[(size_t)i * nc + c];
        for (int c = 0; c < nc; c++) avg[c] += loc[c];
    }
    for (int c = 0; c < nc; c++) avg[c] /= (double)n;
    for (int c = 0; c < nc; c++) nf[c] = 0.0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double loc[MAXC] = {0};
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
            for (int c = 0; c < nc; c++)
            {
                const double t = sumA[i] * avg[c];
                loc[c] += fabs(Apsi[(size_t)i * nc + c] - t) + fabs(source[(size_t)i * nc + c] - t);
            }
        for (int c = 0; c < nc; c++) nf[c] += loc[c];
    }
    for (int c = 0; c < nc; c++) nf[c] = stabilise(nf[c], small_);
    free(sumA);
}

/* TDILUPreconditioner.C:46-79 (rank-local) */
void orc_c_TDILU_calcInvD(const orc_dom* D, double* rD)
{
    for (int cell = 0; cell < D->nCells; cell++) rD[cell] = D->diag[cell];
    for (int face = 0; face < D->nFaces; face++)
        rD[D->u[face]] -= (D->upper[face] * D->lower[face]) * (1.0 / rD[D->l[face]]);
    for (int cell = 0; cell < D->nCells; cell++) rD[cell] = 1.0 / rD[cell];
}

/* TDILUPreconditioner.C:82-125 */
void orc_c_TDILU_precondition(const orc_dom* D, int nc, const double* rD, double* wA, const double* rA)
{
    for (int cell = 0; cell < D->nCells; cell++)
        for (int c = 0; c < nc; c++) wA[cell * nc + c] = rD[cell] * rA[cell * nc + c];
    for (int face = 0; face < D->nFaces; face++)
    {
        const int sface = D->losort[face];
        for (int c = 0; c < nc; c++)
            wA[D->u[sface] * nc + c] -= rD[D->u[sface]] * (D->lower[sface] * wA[D->l[sface] * nc + c]);
    }
    for (int face = D->nFaces - 1; face >= 0; face--)
        for (int c = 0; c < nc; c++)
            wA[D->l[face] * nc + c] -= rD[D->l[face]] * (D->upper[face] * wA[D->u[face] * nc + c]);
}

/* TDILUPreconditioner.C:128-176 */
void orc_c_TDILU_preconditionT(const orc_dom* D, int nc, const double* rD, double* wT, const double* rT)
{
    for (int cell = 0; cell < D->nCells; cell++)
        for (int c = 0; c < nc; c++) wT[cell * nc + c] = rD[cell] * rT[cell * nc + c];
    for (int face = 0; face < D->nFaces; face++)
        for (int c = 0; c < nc; c++)
            wT[D->u[face] * nc + c] -= rD[D->u[face]] * (D->upper[face] * wT[D->l[face] * nc + c]);
    for (int face = D->nFaces - 1; face >= 0; face--)
    {
        const int sface = D->losort[face];
        for (int c = 0; c < nc; c++)
            wT[D->l[sface] * nc + c] -= rD[D->l[sface]] * (D->lower[sface] * wT[D->u[sface] * nc + c]);
    }
}

/* system-level preconditioner apply: none (NoPreconditioner.C:49-56), diagonal (DiagonalPreconditioner.C:40-80),
 * DILU */
void orc_c_precondition(const orc_sys* s, int kind, int nc, double* wAll, const double* rAll, int transpose)
{
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* w = wAll + (size_t)D->cellOffset * nc;
        const double* r = rAll + (size_t)D->cellOffset * nc;
        if (kind == ORC_CPRE_NONE)
            memcpy(w, r, sizeof(double) * (size_t)D->nCells * nc);
        else if (kind == ORC_CPRE_DIAGONAL)
        {
            for (int cell = 0; cell < D->nCells; cell++)
            {
                const double rD = 1.0 / D->diag[cell];
                for (int c = 0; c < nc; c++) w[cell * nc + c] = rD * r[cell * nc + c];
            }
        }
        else
        {
            double* rD = (double*)malloc(sizeof(double) * (D->nCells + 1));
            orc_c_TDILU_calcInvD(D, rD);
            if (transpose) orc_c_TDILU_preconditionT(D, nc, rD, w, r);
            else orc_c_TDILU_precondition(D, nc, rD, w, r);
            free(rD);
        }
    }
}

/* TGaussSeidelSmoother.C:63-153 */
void orc_c_smooth(const orc_sys* s, int nc, double* psiAll, const double* sourceAll, int nSweeps)
{
    const size_t nTot = (size_t)s->nCellsTotal * nc;
    double* bPrimeAll = (double*)malloc(sizeof(double) * (nTot + 1));
    double* snap = (double*)malloc(sizeof(double) * (nTot + 1));
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        /* every rank posts its sends before any rank sweeps: halo = psi at the start of the sweep */
        memcpy(snap, psiAll, sizeof(double) * nTot);
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            double* psi = psiAll + (size_t)D->cellOffset * nc;
            double* bPrime = bPrimeAll + (size_t)D->cellOffset * nc;
            memcpy(bPrime, sourceAll + (size_t)D->cellOffset * nc, sizeof(double) * (size_t)D->nCells * nc);
            c_update_interfaces(s, d, nc, bPrime, snap, 0, -1.0);
            for (int cell = 0; cell < D->nCells; cell++)
            {
                const int fStart = D->ownerStart[cell], fEnd = D->ownerStart[cell + 1];
                const double rD = 1.0 / D->diag[cell];
                for (int c = 0; c < nc; c++)
                {
                    double cur = bPrime[cell * nc + c];
                    for (int f = fStart; f < fEnd; f++) cur -= D->upper[f] * psi[D->u[f] * nc + c];
                    cur = rD * cur;
                    for (int f = fStart; f < fEnd; f++) bPrime[D->u[f] * nc + c] -= D->lower[f] * cur;
                    psi[cell * nc + c] = cur;
                }
            }
        }
    }
    free(bPrimeAll);
    free(snap);
}

/* SolverPerformance.C:60-90; VectorSpaceI.H:661-689 (a comparison holds iff it holds for every component) */
static int c_checkConvergence(orc_cperf* p, const orc_copts* o)
{
    int absOk = 1, relOn = 1, relOk = 1;
    for (int c = 0; c < o->nc; c++)
    {
        absOk = absOk && p->finalResidual[c] < o->tolerance[c];
        relOn = relOn && o->relTol[c] > small_ * 1.0;
        relOk = relOk && p->finalResidual[c] < o->relTol[c] * p->initialResidual[c];
    }
    p->converged = (absOk || (relOn && relOk)) ? 1 : 0;
    return p->converged;
}

/* SolverPerformance.C:32-55 */
static int c_checkSingularity(orc_cperf* p, int nc, const double* wApA)
{
    int all = 1;
    for (int c = 0; c < nc; c++)
    {
        p->singular[c] = wApA[c] < vsmall_;
        all = all && p->singular[c];
    }
    return all;
}

/* PCICG.C:50-184 (bi 0), PBiCICG.C:50-197 (bi 1), PBiCCCG.C:50-192 (bi 2) */
static void c_krylov(const orc_sys* s, const orc_copts* o, double* psi, const double* source, orc_cperf* perf, int bi)
{
    const int nc = o->nc;
    const size_t n = (size_t)s->nCellsTotal, N = n * nc;
    double* pA = (double*)calloc(N + 1, sizeof(double));
    double* wA = (double*)calloc(N + 1, sizeof(double));
    double* rA = (double*)calloc(N + 1, sizeof(double));
    double *pT = NULL, *wT = NULL, *rT = NULL;
    if (bi)
    {
        pT = (double*)calloc(N + 1, sizeof(double));
        wT = (double*)calloc(N + 1, sizeof(double));
        rT = (double*)calloc(N + 1, sizeof(double));
    }
    double wArA[MAXC], wArAold[MAXC], wApA[MAXC], res[MAXC], test[MAXC];
    for (int c = 0; c < nc; c++) wArA[c] = bi == 2 ? 1e15 : great_ * 1.0;

    orc_c_ATmul(s, nc, wA, psi, 0);
    for (size_t i = 0; i < N; i++) rA[i] = source[i] - wA[i];
    if (bi)
    {
        orc_c_ATmul(s, nc, wT, psi, 1);
        for (size_t i = 0; i < N; i++) rT[i] = source[i] - wT[i];
    }
    c_normFactor(s, nc, psi, source, wA, perf->normFactor);
    c_sumCmptMag(s, nc, rA, res);
    for (int c = 0; c < nc; c++) perf->initialResidual[c] = perf->finalResidual[c] = res[c] / perf->normFactor[c];

    if (!c_checkConvergence(perf, o))
    {
        do
        {
            for (int c = 0; c < nc; c++) wArAold[c] = wArA[c];
            orc_c_precondition(s, o->precond, nc, wA, rA, 0);
            if (bi) orc_c_precondition(s, o->precond, nc, wT, rT, 1);
            if (bi == 2)
            {
                const double v = c_sumProd(s, nc, o->ipw, wA, rT);
                for (int c = 0; c < nc; c++) wArA[c] = v;
            }
            else c_sumCmptProd(s, nc, wA, bi ? rT : rA, wArA);

            if (perf->nIterations == 0)
            {
                memcpy(pA, wA, sizeof(double) * N);
                if (bi) memcpy(pT, wT, sizeof(double) * N);
            }
            else
            {
                double beta[MAXC];
                for (int c = 0; c < nc; c++)
                    beta[c] = bi == 2 ? wArA[c] / wArAold[c] : wArA[c] / stabilise(wArAold[c], vsmall_);
                for (size_t i = 0; i < n; i++)
                    for (int c = 0; c < nc; c++)
                    {
                        pA[i * nc + c] = wA[i * nc + c] + beta[c] * pA[i * nc + c];
                        if (bi) pT[i * nc + c] = wT[i * nc + c] + beta[c] * pT[i * nc + c];
                    }
            }
            orc_c_ATmul(s, nc, wA, pA, 0);
            if (bi) orc_c_ATmul(s, nc, wT, pT, 1);
            if (bi == 2)
            {
                const double v = c_sumProd(s, nc, o->ipw, wA, pT);
                for (int c = 0; c < nc; c++) wApA[c] = v;
            }
            else c_sumCmptProd(s, nc, wA, bi ? pT : pA, wApA);

            for (int c = 0; c < nc; c++) test[c] = fabs(wApA[c]) / perf->normFactor[c];
            if (c_checkSingularity(perf, nc, test)) break;

            double alpha[MAXC];
            for (int c = 0; c < nc; c++)
                alpha[c] = bi == 2 ? wArA[c] / wApA[c] : wArA[c] / stabilise(wApA[c], vsmall_);
            for (size_t i = 0; i < n; i++)
                for (int c = 0; c < nc; c++)
                {
                    psi[i * nc + c] += alpha[c] * pA[i * nc + c];
                    rA[i * nc + c] -= alpha[c] * wA[i * nc + c];
                    if (bi) rT[i * nc + c] -= alpha[c] * wT[i * nc + c];
                }
            c_sumCmptMag(s, nc, rA, res);
            for (int c = 0; c < nc; c++) perf->finalResidual[c] = res[c] / perf->normFactor[c];
        } while (perf->nIterations++ < o->maxIter && !c_checkConvergence(perf, o));
    }
    free(pA); free(wA); free(rA);
    if (bi) { free(pT); free(wT); free(rT); }
}

/* SmoothSolver.C:61-151 */
static void c_smoothSolver(const orc_sys* s, const orc_copts* o, double* psi, const double* source, orc_cperf* perf)
{
    const int nc = o->nc;
    const size_t N = (size_t)s->nCellsTotal * nc;
    if (o->nSweeps < 0)
    {
        orc_c_smooth(s, nc, psi, source, -o->nSweeps);
        perf->nIterations -= o->nSweeps;
        return;
    }
    double* Apsi = (double*)calloc(N + 1, sizeof(double));
    double* rA = (double*)calloc(N + 1, sizeof(double));
    double res[MAXC];
    orc_c_ATmul(s, nc, Apsi, psi, 0);
    c_normFactor(s, nc, psi, source, Apsi, perf->normFactor);
    for (size_t i = 0; i < N; i++) rA[i] = source[i] - Apsi[i];
    c_sumCmptMag(s, nc, rA, res);
    for (int c = 0; c < nc; c++) perf->initialResidual[c] = perf->finalResidual[c] = res[c] / perf->normFactor[c];
    if (!c_checkConvergence(perf, o))
    {
        do
        {
            orc_c_smooth(s, nc, psi, source, o->nSweeps);
            orc_c_residual(s, nc, rA, psi, source);
            c_sumCmptMag(s, nc, rA, res);
            for (int c = 0; c < nc; c++) perf->finalResidual[c] = res[c] / perf->normFactor[c];
        } while ((perf->nIterations += o->nSweeps) < o->maxIter && !c_checkConvergence(perf, o));
    }
    free(Apsi); free(rA);
}

/* LduMatrixSolver.C:33-111: returns 0, or -16 where the reference raises FatalIOError (name not in the table) */
int orc_c_solve(const orc_sys* s, const orc_copts* o, double* psi, const double* source, orc_cperf* perf)
{
    memset(perf, 0, sizeof(*perf));
    int anyFaces = 0, sym = 1;
    for (int d = 0; d < s->nDom; d++)
    {
        if (s->dom[d].nFaces) anyFaces = 1;
        if (s->dom[d].lower != s->dom[d].upper) sym = 0;
    }
    if (!anyFaces)
    {
        /* DiagonalSolver.C:56-76 */
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            for (int cell = 0; cell < D->nCells; cell++)
                for (int c = 0; c < o->nc; c++)
                {
                    const size_t i = (size_t)(D->cellOffset + cell) * o->nc + c;
                    psi[i] = source[i] / D->diag[cell];
                }
        }
        perf->converged = 1;
        return 0;
    }
    if (sym && (o->solver == ORC_C_PBICCCG || o->solver == ORC_C_PBICICG)) return -16;
    if (!sym && o->solver == ORC_C_PCICG) return -16;
    if (o->solver != ORC_C_SMOOTH && sym && o->precond == ORC_CPRE_DILU) return -16;
    switch (o->solver)
    {
    case ORC_C_PCICG: c_krylov(s, o, psi, source, perf, 0); break;
    case ORC_C_PBICICG: c_krylov(s, o, psi, source, perf, 1); break;
    case ORC_C_PBICCCG: c_krylov(s, o, psi, source, perf, 2); break;
    case ORC_C_SMOOTH: c_smoothSolver(s, o, psi, source, perf); break;
    default: return -3;
    }
    return 0;
}
