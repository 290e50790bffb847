/*
 * TEST INFRASTRUCTURE - CPU oracle, part 1: addressing, matrix ops, reductions,
 * preconditioners, smoothers, PCG / PBiCG / smoothSolver / diagonalSolver.
 * See ldu_oracle.h.  Reference paths are relative to
 * /root/reference/src/OpenFOAM/matrices/lduMatrix/ unless stated otherwise.
 */
#include "ldu_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* SolverPerformance.H:269-275 */
#define ORC_GREAT 1e20
#define ORC_SMALL 1e-20
#define ORC_VSMALL 1e-300

void orc_default_opts(orc_opts* o)
{
    memset(o, 0, sizeof(*o));
    o->solver = ORC_PCG;
    o->precond = ORC_PRE_DIC;
    o->smoother = ORC_SM_GS;
    o->tolerance = 1e-6;              /* lduMatrixSolver.C:164-169 */
    o->relTol = 0;
    o->maxIter = 1000;
    o->nSweeps = 1;                   /* smoothSolver.C:73 */
    o->nPreSweeps = 0;                /* GAMGSolver.C:66-76 */
    o->preSweepsLevelMultiplier = 1;
    o->maxPreSweeps = 4;
    o->nPostSweeps = 2;
    o->postSweepsLevelMultiplier = 1;
    o->maxPostSweeps = 4;
    o->nFinestSweeps = 2;
    o->interpolateCorrection = 0;
    o->scaleCorrection = -1;
    o->nCellsInCoarsestLevel = 10;
    o->mergeLevels = 1;
    o->agglomerator = ORC_AGG_FACEAREAPAIR;
    o->nVcycles = 2;                  /* GAMGPreconditioner.C:60 */
    o->directSolveCoarsest = 0;       /* GAMGSolver.C:76 */
}

/* ------------------------------------------------------------------ addressing */

/* lduAddressing/lduAddressing.C:31-89: faces bucketed by upper cell, stable */
void orc_calc_losort(int nCells, int nFaces, const int* u, int* losort)
{
    int* start = (int*)calloc((size_t)nCells + 1, sizeof(int));
    for (int f = 0; f < nFaces; f++) start[u[f] + 1]++;
    for (int c = 0; c < nCells; c++) start[c + 1] += start[c];
    for (int f = 0; f < nFaces; f++) losort[start[u[f]]++] = f;
    free(start);
}

/* lduAddressing.C:92-126 */
void orc_calc_ownerStart(int nCells, int nFaces, const int* own, int* ownStart)
{
    for (int i = 0; i <= nCells; i++) ownStart[i] = nFaces;
    ownStart[0] = 0;
    int nOwnStart = 0;
    int i = 1;
    for (int faceI = 0; faceI < nFaces; faceI++)
    {
        int curOwn = own[faceI];
        if (curOwn > nOwnStart)
        {
            while (i <= curOwn) ownStart[i++] = faceI;
            nOwnStart = curOwn;
        }
    }
}

/* lduAddressing.C:129-169 */
void orc_calc_losortStart(int nCells, int nFaces, const int* nbr, const int* lsrt, int* lsrtStart)
{
    for (int i = 0; i <= nCells; i++) lsrtStart[i] = 0;
    int nLsrtStart = 0;
    int i = 0;
    for (int faceI = 0; faceI < nFaces; faceI++)
    {
        const int curNbr = nbr[lsrt[faceI]];
        if (curNbr > nLsrtStart)
        {
            while (i <= curNbr) lsrtStart[i++] = faceI;
            nLsrtStart = curNbr;
        }
    }
    /* set up last lookup by hand (entries of cells beyond the last neighbour
     * keep their initial 0, literally as in the reference) */
    lsrtStart[nCells] = nFaces;
}

void orc_sys_finalize(orc_sys* s)
{
    int off = 0;
    for (int d = 0; d < s->nDom; d++)
    {
        orc_dom* D = &s->dom[d];
        D->cellOffset = off;
        off += D->nCells;
        D->losort = (int*)malloc(sizeof(int) * (size_t)(D->nFaces > 0 ? D->nFaces : 1));
        D->ownerStart = (int*)malloc(sizeof(int) * ((size_t)D->nCells + 1));
        D->losortStart = (int*)malloc(sizeof(int) * ((size_t)D->nCells + 1));
        orc_calc_losort(D->nCells, D->nFaces, D->u, D->losort);
        orc_calc_ownerStart(D->nCells, D->nFaces, D->l, D->ownerStart);
        orc_calc_losortStart(D->nCells, D->nFaces, D->u, D->losort, D->losortStart);
    }
    s->nCellsTotal = off;
}

void orc_sys_free_derived(orc_sys* s)
{
    for (int d = 0; d < s->nDom; d++)
    {
        free(s->dom[d].losort); s->dom[d].losort = 0;
        free(s->dom[d].ownerStart); s->dom[d].ownerStart = 0;
        free(s->dom[d].losortStart); s->dom[d].losortStart = 0;
    }
}

/* ------------------------------------------------------------------ halo emulation
 * processorFvPatchScalarField.C:36-144 : the sender gathers psi[faceCells]
 * (patchInternalField) in initInterfaceMatrixUpdate; the receiver applies
 *    result[faceCells[i]] -= coeffs[i]*pnf[i]          (:125-128)
 * in updateInterfaceMatrix.  `useInt` selects interfaceIntCoeffs (Tmul),
 * `sign` = -1 reproduces the negated coefficient copies of residual / smoothers
 * (lduMatrixATmul.C:225-244, GaussSeidelSmoother.C:110-122).
 */
static void update_interfaces(const orc_sys* s, int d, double* result /* domain-local */,
                              const double* psiAll, int useInt, double sign)
{
    const orc_dom* D = &s->dom[d];
    for (int p = 0; p < D->nPatches; p++)
    {
        const orc_patch* P = &D->patches[p];
        const orc_dom* N = &s->dom[P->nbrDom];
        const orc_patch* NP = &N->patches[P->nbrPatch];
        const double* coeffs = useInt ? P->intCoeffs : P->bouCoeffs;
        const double* psiN = psiAll + N->cellOffset;
        for (int i = 0; i < P->n; i++)
        {
            const double c = sign < 0 ? -coeffs[i] : coeffs[i];
            result[P->faceCells[i]] -= c * psiN[NP->faceCells[i]];
        }
    }
}

/* ------------------------------------------------------------------ matrix ops */

/* lduMatrix/lduMatrixATmul.C:34-92 */
void orc_Amul(const orc_sys* s, double* ApsiAll, const double* psiAll)
{
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* Apsi = ApsiAll + D->cellOffset;
        const double* psi = psiAll + D->cellOffset;
        for (int cell = 0; cell < D->nCells; cell++) Apsi[cell] = D->diag[cell] * psi[cell];
        for (int face = 0; face < D->nFaces; face++)
        {
            Apsi[D->u[face]] += D->lower[face] * psi[D->l[face]];
            Apsi[D->l[face]] += D->upper[face] * psi[D->u[face]];
        }
        update_interfaces(s, d, Apsi, psiAll, 0, 1.0);
    }
}

/* lduMatrixATmul.C:95-151 */
void orc_Tmul(const orc_sys* s, double* TpsiAll, const double* psiAll)
{
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* Tpsi = TpsiAll + D->cellOffset;
        const double* psi = psiAll + D->cellOffset;
        for (int cell = 0; cell < D->nCells; cell++) Tpsi[cell] = D->diag[cell] * psi[cell];
        for (int face = 0; face < D->nFaces; face++)
        {
            Tpsi[D->u[face]] += D->upper[face] * psi[D->l[face]];
            Tpsi[D->l[face]] += D->lower[face] * psi[D->u[face]];
        }
        update_interfaces(s, d, Tpsi, psiAll, 1, 1.0);
    }
}

/* lduMatrixATmul.C:154-200 */
void orc_sumA(const orc_sys* s, double* sumAAll)
{
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* sumA = sumAAll + D->cellOffset;
        for (int cell = 0; cell < D->nCells; cell++) sumA[cell] = D->diag[cell];
        for (int face = 0; face < D->nFaces; face++)
        {
            sumA[D->u[face]] += D->lower[face];
            sumA[D->l[face]] += D->upper[face];
        }
        for (int p = 0; p < D->nPatches; p++)
        {
            const orc_patch* P = &D->patches[p];
            for (int i = 0; i < P->n; i++) sumA[P->faceCells[i]] -= P->bouCoeffs[i];
        }
    }
}

/* lduMatrixATmul.C:203-280 */
void orc_residual(const orc_sys* s, double* rAAll, const double* psiAll, const double* sourceAll)
{
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* rA = rAAll + D->cellOffset;
        const double* psi = psiAll + D->cellOffset;
        const double* source = sourceAll + D->cellOffset;
        for (int cell = 0; cell < D->nCells; cell++)
            rA[cell] = source[cell] - D->diag[cell] * psi[cell];
        for (int face = 0; face < D->nFaces; face++)
        {
            rA[D->u[face]] -= D->lower[face] * psi[D->l[face]];
            rA[D->l[face]] -= D->upper[face] * psi[D->u[face]];
        }
        update_interfaces(s, d, rA, psiAll, 0, -1.0);
    }
}

/* lduMatrixTemplates.C:34-65 (H), lduMatrixATmul.C:298-327 (H1), lduMatrixTemplates.C:78-110 (faceH) */
void orc_H(const orc_dom* D, double* H, const double* psi)
{
    for (int c = 0; c < D->nCells; c++) H[c] = 0;
    for (int face = 0; face < D->nFaces; face++)
    {
        H[D->u[face]] -= D->lower[face] * psi[D->l[face]];
        H[D->l[face]] -= D->upper[face] * psi[D->u[face]];
    }
}
void orc_H1(const orc_dom* D, double* H1)
{
    for (int c = 0; c < D->nCells; c++) H1[c] = 0;
    for (int face = 0; face < D->nFaces; face++)
    {
        H1[D->u[face]] -= D->lower[face];
        H1[D->l[face]] -= D->upper[face];
    }
}
void orc_faceH(const orc_dom* D, double* faceH, const double* psi)
{
    for (int face = 0; face < D->nFaces; face++)
        faceH[face] = D->upper[face] * psi[D->u[face]] - D->lower[face] * psi[D->l[face]];
}

/* ------------------------------------------------------------------ reductions
 * src/OpenFOAM/fields/Fields/scalarField/scalarField.C:91-104 (sumProd),
 * Fields/Field/FieldFunctions.C:421-434 (sumMag), :477-503 (g* = local + reduce) */
/* the local sums of the ranks (one thread each in the OpenMP build), then the reduce in rank order */
static double sum_in_rank_order(const double* part, int n)
{
    double g = 0;
    for (int d = 0; d < n; d++) g = (d == 0) ? part[d] : g + part[d];
    return g;
}

double orc_gSumProd(const orc_sys* s, const double* a, const double* b)
{
    double part[ORC_MAXDOM];
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        const double* x = a + D->cellOffset;
        const double* y = b + D->cellOffset;
        double sum = 0.0;
        for (int i = 0; i < D->nCells; i++) sum += x[i] * y[i];
        part[d] = sum;
    }
    return sum_in_rank_order(part, s->nDom);
}

double orc_gSumMag(const orc_sys* s, const double* a)
{
    double part[ORC_MAXDOM];
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        const double* x = a + D->cellOffset;
        double sum = 0.0;
        for (int i = 0; i < D->nCells; i++) sum += fabs(x[i]);
        part[d] = sum;
    }
    return sum_in_rank_order(part, s->nDom);
}

/* lduMatrix/lduMatrixSolver.C:179-197; gAverage FieldFunctions.C:514-533 */
double orc_normFactor(const orc_sys* s, const double* psi, const double* source,
                      const double* Apsi, double* tmp)
{
    orc_sumA(s, tmp);
    double gs = 0;
    long n = 0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double sum = 0.0;
        for (int i = 0; i < D->nCells; i++) sum += psi[D->cellOffset + i];
        gs = (d == 0) ? sum : gs + sum;
        n += D->nCells;
    }
    const double avg = n > 0 ? gs / (double)n : 0.0;
    for (int i = 0; i < s->nCellsTotal; i++) tmp[i] *= avg;

    double g = 0;
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double sum = 0.0;
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
            sum += fabs(Apsi[i] - tmp[i]) + fabs(source[i] - tmp[i]);
        g = (d == 0) ? sum : g + sum;
    }
    return g + ORC_SMALL;
}

/* ------------------------------------------------------------------ preconditioners */

/* preconditioners/DICPreconditioner/DICPreconditioner.C:57-84 */
void orc_DIC_calcReciprocalD(const orc_dom* D, double* rD)
{
    for (int c = 0; c < D->nCells; c++) rD[c] = D->diag[c];
    for (int face = 0; face < D->nFaces; face++)
        rD[D->u[face]] -= D->upper[face] * D->upper[face] / rD[D->l[face]];
    for (int c = 0; c < D->nCells; c++) rD[c] = 1.0 / rD[c];
}

/* preconditioners/DILUPreconditioner/DILUPreconditioner.C:57-85 */
void orc_DILU_calcReciprocalD(const orc_dom* D, double* rD)
{
    for (int c = 0; c < D->nCells; c++) rD[c] = D->diag[c];
    for (int face = 0; face < D->nFaces; face++)
        rD[D->u[face]] -= D->upper[face] * D->lower[face] / rD[D->l[face]];
    for (int c = 0; c < D->nCells; c++) rD[c] = 1.0 / rD[c];
}

/* DICPreconditioner.C:87-123 */
void orc_DIC_precondition(const orc_dom* D, const double* rD, double* wA, const double* rA)
{
    for (int c = 0; c < D->nCells; c++) wA[c] = rD[c] * rA[c];
    for (int face = 0; face < D->nFaces; face++)
        wA[D->u[face]] -= rD[D->u[face]] * D->upper[face] * wA[D->l[face]];
    for (int face = D->nFaces - 1; face >= 0; face--)
        wA[D->l[face]] -= rD[D->l[face]] * D->upper[face] * wA[D->u[face]];
}

/* preconditioners/FDICPreconditioner/FDICPreconditioner.C:87-123 (setup :42-84 by the caller) */
void orc_FDIC_precondition(const orc_dom* D, const double* rD, const double* rDuUpper,
                           const double* rDlUpper, double* wA, const double* rA)
{
    for (int c = 0; c < D->nCells; c++) wA[c] = rD[c] * rA[c];
    for (int face = 0; face < D->nFaces; face++)
        wA[D->u[face]] -= rDuUpper[face] * wA[D->l[face]];
    for (int face = D->nFaces - 1; face >= 0; face--)
        wA[D->l[face]] -= rDlUpper[face] * wA[D->u[face]];
}

/* DILUPreconditioner.C:88-135 */
void orc_DILU_precondition(const orc_dom* D, const double* rD, double* wA, const double* rA)
{
    for (int c = 0; c < D->nCells; c++) wA[c] = rD[c] * rA[c];
    for (int face = 0; face < D->nFaces; face++)
    {
        const int sface = D->losort[face];
        wA[D->u[sface]] -= rD[D->u[sface]] * D->lower[sface] * wA[D->l[sface]];
    }
    for (int face = D->nFaces - 1; face >= 0; face--)
        wA[D->l[face]] -= rD[D->l[face]] * D->upper[face] * wA[D->u[face]];
}

/* DILUPreconditioner.C:138-185 */
void orc_DILU_preconditionT(const orc_dom* D, const double* rD, double* wT, const double* rT)
{
    for (int c = 0; c < D->nCells; c++) wT[c] = rD[c] * rT[c];
    for (int face = 0; face < D->nFaces; face++)
        wT[D->u[face]] -= rD[D->u[face]] * D->upper[face] * wT[D->l[face]];
    for (int face = D->nFaces - 1; face >= 0; face--)
    {
        const int sface = D->losort[face];
        wT[D->l[sface]] -= rD[D->l[sface]] * D->lower[sface] * wT[D->u[sface]];
    }
}

/* A preconditioner object bound to a system (rank-local factors per domain). */
typedef struct precond_t {
    int kind;
    double* rD;        /* concatenated */
    double** rDuUpper; /* FDIC per domain */
    double** rDlUpper;
} precond_t;

static precond_t* precond_new(const orc_sys* s, int kind)
{
    precond_t* P = (precond_t*)calloc(1, sizeof(precond_t));
    P->kind = kind;
    P->rD = (double*)malloc(sizeof(double) * (size_t)(s->nCellsTotal > 0 ? s->nCellsTotal : 1));
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* rD = P->rD + D->cellOffset;
        switch (kind)
        {
        case ORC_PRE_DIAGONAL:
            /* diagonalPreconditioner.C:46-69: rD = 1/diag */
            for (int c = 0; c < D->nCells; c++) rD[c] = 1.0 / D->diag[c];
            break;
        case ORC_PRE_DIC:
            orc_DIC_calcReciprocalD(D, rD);
            break;
        case ORC_PRE_DILU:
            orc_DILU_calcReciprocalD(D, rD);
            break;
        case ORC_PRE_FDIC:
        {
            /* FDICPreconditioner.C:42-84 */
            if (!P->rDuUpper)
            {
                P->rDuUpper = (double**)calloc((size_t)s->nDom, sizeof(double*));
                P->rDlUpper = (double**)calloc((size_t)s->nDom, sizeof(double*));
            }
            for (int c = 0; c < D->nCells; c++) rD[c] = D->diag[c];
            for (int face = 0; face < D->nFaces; face++)
                rD[D->u[face]] -= (D->upper[face] * D->upper[face]) / rD[D->l[face]];
            for (int c = 0; c < D->nCells; c++) rD[c] = 1.0 / rD[c];
            P->rDuUpper[d] = (double*)malloc(sizeof(double) * (size_t)(D->nFaces + 1));
            P->rDlUpper[d] = (double*)malloc(sizeof(double) * (size_t)(D->nFaces + 1));
            for (int face = 0; face < D->nFaces; face++)
            {
                P->rDuUpper[d][face] = rD[D->u[face]] * D->upper[face];
                P->rDlUpper[d][face] = rD[D->l[face]] * D->upper[face];
            }
            break;
        }
        default:
            break;
        }
    }
    return P;
}

static void precond_free(const orc_sys* s, precond_t* P)
{
    if (!P) return;
    if (P->rDuUpper)
    {
        for (int d = 0; d < s->nDom; d++) { free(P->rDuUpper[d]); free(P->rDlUpper[d]); }
        free(P->rDuUpper); free(P->rDlUpper);
    }
    free(P->rD);
    free(P);
}

static void precond_apply(const orc_sys* s, const precond_t* P, double* wAll, const double* rAll,
                          int transpose)
{
    ORC_PAR
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        double* w = wAll + D->cellOffset;
        const double* r = rAll + D->cellOffset;
        const double* rD = P->rD + D->cellOffset;
        switch (P->kind)
        {
        case ORC_PRE_NONE:      /* noPreconditioner.C:58-74 (and its preconditionT) */
            for (int c = 0; c < D->nCells; c++) w[c] = r[c];
            break;
        case ORC_PRE_DIAGONAL:  /* diagonalPreconditioner.C:72-87: wA = rD*rA */
            for (int c = 0; c < D->nCells; c++) w[c] = rD[c] * r[c];
            break;
        case ORC_PRE_DIC:
            orc_DIC_precondition(D, rD, w, r);
            break;
        case ORC_PRE_FDIC:
            orc_FDIC_precondition(D, rD, P->rDuUpper[d], P->rDlUpper[d], w, r);
            break;
        case ORC_PRE_DILU:
            if (transpose) orc_DILU_preconditionT(D, rD, w, r);
            else orc_DILU_precondition(D, rD, w, r);
            break;
        default:
            break;
        }
    }
}

/* ------------------------------------------------------------------ smoothers */

/* One GaussSeidel forward sweep of one domain given bPrime (already holding
 * source + coupled contributions): smoothers/GaussSeidel/GaussSeidelSmoother.C:147-176 */
static void gs_forward(const orc_dom* D, double* psi, double* bPrime)
{
    int fStart;
    int fEnd = D->ownerStart[0];
    for (int celli = 0; celli < D->nCells; celli++)
    {
        fStart = fEnd;
        fEnd = D->ownerStart[celli + 1];
        double psii = bPrime[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            psii -= D->upper[facei] * psi[D->u[facei]];
        psii /= D->diag[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            bPrime[D->u[facei]] -= D->lower[facei] * psii;
        psi[celli] = psii;
    }
}

/* cells [c0, c1) of the same loop (nonBlockingGaussSeidelSmoother.C:167-196 and :207-235) */
static void gs_forward_range(const orc_dom* D, double* psi, double* bPrime, int c0, int c1)
{
    for (int celli = c0; celli < c1; celli++)
    {
        const int fStart = D->ownerStart[celli], fEnd = D->ownerStart[celli + 1];
        double psii = bPrime[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            psii -= D->upper[facei] * psi[D->u[facei]];
        psii /= D->diag[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            bPrime[D->u[facei]] -= D->lower[facei] * psii;
        psi[celli] = psii;
    }
}

/* smoothers/nonBlockingGaussSeidel/nonBlockingGaussSeidelSmoother.C:46-245: the cells below
 * blockStart_ (= smallest cell index touched by a coupled patch, :66-79) are swept BEFORE the coupled
 * contributions are added to bPrime (:198-205), the rest after: same values as GaussSeidel, but a
 * boundary cell accumulates source, lower neighbours < blockStart, interfaces, lower neighbours >= blockStart */
static void smooth_gs_nonblocking(const orc_sys* s, double* psiAll, const double* sourceAll, int nSweeps)
{
    double* bPrimeAll = (double*)malloc(sizeof(double) * (size_t)(s->nCellsTotal + 1));
    int* blockStart = (int*)malloc(sizeof(int) * (size_t)(s->nDom + 1));
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        blockStart[d] = D->nCells;
        for (int p = 0; p < D->nPatches; p++)
            for (int i = 0; i < D->patches[p].n; i++)
                if (D->patches[p].faceCells[i] < blockStart[d]) blockStart[d] = D->patches[p].faceCells[i];
    }
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            double* bPrime = bPrimeAll + D->cellOffset;
            for (int c = 0; c < D->nCells; c++) bPrime[c] = sourceAll[D->cellOffset + c];
            gs_forward_range(D, psiAll + D->cellOffset, bPrime, 0, blockStart[d]);
        }
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
            update_interfaces(s, d, bPrimeAll + s->dom[d].cellOffset, psiAll, 0, -1.0);
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            gs_forward_range(D, psiAll + D->cellOffset, bPrimeAll + D->cellOffset, blockStart[d], D->nCells);
        }
    }
    free(blockStart);
    free(bPrimeAll);
}

/* smoothers/symGaussSeidel/symGaussSeidelSmoother.C:178-205 (bPrime NOT reset) */
static void gs_reverse(const orc_dom* D, double* psi, double* bPrime)
{
    int fEnd;
    int fStart = D->ownerStart[D->nCells];
    for (int celli = D->nCells - 1; celli >= 0; celli--)
    {
        fEnd = fStart;
        fStart = D->ownerStart[celli];
        double psii = bPrime[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            psii -= D->upper[facei] * psi[D->u[facei]];
        psii /= D->diag[celli];
        for (int facei = fStart; facei < fEnd; facei++)
            bPrime[D->u[facei]] -= D->lower[facei] * psii;
        psi[celli] = psii;
    }
}

static void smooth_gs(const orc_sys* s, double* psiAll, const double* sourceAll, int nSweeps, int sym)
{
    double* bPrimeAll = (double*)malloc(sizeof(double) * (size_t)(s->nCellsTotal + 1));
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        /* phase 1 (all ranks, before any rank sweeps): bPrime = source, then the
         * coupled boundary added Jacobi-style with negated coefficients
         * (GaussSeidelSmoother.C:98-145) */
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            double* bPrime = bPrimeAll + D->cellOffset;
            for (int c = 0; c < D->nCells; c++) bPrime[c] = sourceAll[D->cellOffset + c];
            update_interfaces(s, d, bPrime, psiAll, 0, -1.0);
        }
        /* phase 2: rank-local sweeps */
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            gs_forward(D, psiAll + D->cellOffset, bPrimeAll + D->cellOffset);
            if (sym) gs_reverse(D, psiAll + D->cellOffset, bPrimeAll + D->cellOffset);
        }
    }
    free(bPrimeAll);
}

/* smoothers/DIC/DICSmoother.C:67-116, DILU/DILUSmoother.C:67-119 (plain face order,
 * not losort), FDIC/FDICSmoother.C:98-146 */
static void smooth_dic(const orc_sys* s, int kind, double* psiAll, const double* sourceAll, int nSweeps)
{
    precond_t* P = precond_new(s, kind == ORC_SM_DILU ? ORC_PRE_DILU
                                  : kind == ORC_SM_FDIC ? ORC_PRE_FDIC : ORC_PRE_DIC);
    double* rAAll = (double*)malloc(sizeof(double) * (size_t)(s->nCellsTotal + 1));
    for (int sweep = 0; sweep < nSweeps; sweep++)
    {
        orc_residual(s, rAAll, psiAll, sourceAll);
        ORC_PAR
        for (int d = 0; d < s->nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            double* rA = rAAll + D->cellOffset;
            const double* rD = P->rD + D->cellOffset;
            for (int c = 0; c < D->nCells; c++) rA[c] *= rD[c];
            if (kind == ORC_SM_FDIC)
            {
                for (int face = 0; face < D->nFaces; face++)
                    rA[D->u[face]] -= P->rDuUpper[d][face] * rA[D->l[face]];
                for (int face = D->nFaces - 1; face >= 0; face--)
                    rA[D->l[face]] -= P->rDlUpper[d][face] * rA[D->u[face]];
            }
            else
            {
                const double* lo = (kind == ORC_SM_DILU) ? D->lower : D->upper;
                for (int face = 0; face < D->nFaces; face++)
                {
                    const int u = D->u[face];
                    rA[u] -= rD[u] * lo[face] * rA[D->l[face]];
                }
                for (int face = D->nFaces - 1; face >= 0; face--)
                {
                    const int l = D->l[face];
                    rA[l] -= rD[l] * D->upper[face] * rA[D->u[face]];
                }
            }
            double* psi = psiAll + D->cellOffset;
            for (int c = 0; c < D->nCells; c++) psi[c] += rA[c];
        }
    }
    free(rAAll);
    precond_free(s, P);
}

void orc_smooth(const orc_sys* s, int smoother, double* psi, const double* source, int nSweeps)
{
    switch (smoother)
    {
    case ORC_SM_GS:    smooth_gs(s, psi, source, nSweeps, 0); break;
    case ORC_SM_SYMGS: smooth_gs(s, psi, source, nSweeps, 1); break;
    case ORC_SM_NONBLOCKINGGS: smooth_gs_nonblocking(s, psi, source, nSweeps); break;
    case ORC_SM_DIC:
    case ORC_SM_DILU:
    case ORC_SM_FDIC:  smooth_dic(s, smoother, psi, source, nSweeps); break;
    case ORC_SM_DICGS: /* smoothers/DICGaussSeidel/DICGaussSeidelSmoother.C:79-89 */
        smooth_dic(s, ORC_SM_DIC, psi, source, nSweeps);
        smooth_gs(s, psi, source, nSweeps, 0);
        break;
    case ORC_SM_DILUGS:
        smooth_dic(s, ORC_SM_DILU, psi, source, nSweeps);
        smooth_gs(s, psi, source, nSweeps, 0);
        break;
    default: break;
    }
}

/* ------------------------------------------------------------------ convergence */

/* src/OpenFOAM/matrices/LduMatrix/LduMatrix/SolverPerformance.C:59-91 */
static int check_convergence(orc_perf* p, double tol, double relTol)
{
    if (p->finalResidual < tol
        || (relTol > ORC_SMALL && p->finalResidual < relTol * p->initialResidual))
        p->converged = 1;
    else
        p->converged = 0;
    return p->converged;
}

/* SolverPerformance.C:31-44 */
static int check_singularity(orc_perf* p, double wApA)
{
    p->singular = wApA < ORC_VSMALL;
    return p->singular;
}

static void hist_push(orc_perf* p, double* hist)
{
    if (hist) hist[p->nHist] = p->finalResidual;
    p->nHist++;
}

/* forward declarations from ldu_oracle_gamg.c */
orc_perf orc_gamg_solve(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                        const double* faceWeights, double* resHist);
typedef struct orc_gamg_pre orc_gamg_pre;
orc_gamg_pre* orc_gamg_pre_new(const orc_sys* s, const orc_opts* o, const double* faceWeights);
void orc_gamg_pre_apply(orc_gamg_pre* g, double* wA, const double* rA);
void orc_gamg_pre_free(orc_gamg_pre* g);

/* ------------------------------------------------------------------ PCG
 * solvers/PCG/PCG.C:65-182 */
static orc_perf solve_pcg(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                          const double* faceWeights, double* hist)
{
    orc_perf perf; memset(&perf, 0, sizeof(perf));
    const int n = s->nCellsTotal;
    double* pA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* wA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* rA = (double*)malloc(sizeof(double) * (size_t)(n + 1));

    double wArA = ORC_GREAT;
    double wArAold = wArA;

    orc_Amul(s, wA, psi);
    for (int i = 0; i < n; i++) rA[i] = source[i] - wA[i];
    const double normFactor = orc_normFactor(s, psi, source, wA, pA);
    perf.normFactor = normFactor;
    perf.initialResidual = orc_gSumMag(s, rA) / normFactor;
    perf.finalResidual = perf.initialResidual;
    hist_push(&perf, hist);

    if (!check_convergence(&perf, o->tolerance, o->relTol))
    {
        precond_t* P = 0;
        orc_gamg_pre* G = 0;
        if (o->precond == ORC_PRE_GAMG) G = orc_gamg_pre_new(s, o, faceWeights);
        else P = precond_new(s, o->precond);
        do
        {
            wArAold = wArA;
            if (G) orc_gamg_pre_apply(G, wA, rA);
            else precond_apply(s, P, wA, rA, 0);
            wArA = orc_gSumProd(s, wA, rA);
            if (perf.nIterations == 0)
            {
                for (int c = 0; c < n; c++) pA[c] = wA[c];
            }
            else
            {
                const double beta = wArA / wArAold;
                for (int c = 0; c < n; c++) pA[c] = wA[c] + beta * pA[c];
            }
            orc_Amul(s, wA, pA);
            const double wApA = orc_gSumProd(s, wA, pA);
            if (check_singularity(&perf, fabs(wApA) / normFactor)) break;
            const double alpha = wArA / wApA;
            for (int c = 0; c < n; c++)
            {
                psi[c] += alpha * pA[c];
                rA[c] -= alpha * wA[c];
            }
            perf.finalResidual = orc_gSumMag(s, rA) / normFactor;
            hist_push(&perf, hist);
        } while (perf.nIterations++ < o->maxIter
                 && !check_convergence(&perf, o->tolerance, o->relTol));
        if (G) orc_gamg_pre_free(G);
        precond_free(s, P);
    }
    free(pA); free(wA); free(rA);
    return perf;
}

/* ------------------------------------------------------------------ PBiCG
 * solvers/PBiCG/PBiCG.C:65-198 */
static orc_perf solve_pbicg(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                            double* hist)
{
    orc_perf perf; memset(&perf, 0, sizeof(perf));
    const int n = s->nCellsTotal;
    double* pA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* pT = (double*)calloc((size_t)(n + 1), sizeof(double));
    double* wA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* wT = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* rA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* rT = (double*)malloc(sizeof(double) * (size_t)(n + 1));

    double wArT = ORC_GREAT;
    double wArTold = wArT;

    orc_Amul(s, wA, psi);
    orc_Tmul(s, wT, psi);
    for (int i = 0; i < n; i++) rA[i] = source[i] - wA[i];
    for (int i = 0; i < n; i++) rT[i] = source[i] - wT[i];
    const double normFactor = orc_normFactor(s, psi, source, wA, pA);
    perf.normFactor = normFactor;
    perf.initialResidual = orc_gSumMag(s, rA) / normFactor;
    perf.finalResidual = perf.initialResidual;
    hist_push(&perf, hist);

    if (!check_convergence(&perf, o->tolerance, o->relTol))
    {
        precond_t* P = precond_new(s, o->precond);
        do
        {
            wArTold = wArT;
            precond_apply(s, P, wA, rA, 0);
            precond_apply(s, P, wT, rT, 1);
            wArT = orc_gSumProd(s, wA, rT);
            if (perf.nIterations == 0)
            {
                for (int c = 0; c < n; c++) { pA[c] = wA[c]; pT[c] = wT[c]; }
            }
            else
            {
                const double beta = wArT / wArTold;
                for (int c = 0; c < n; c++)
                {
                    pA[c] = wA[c] + beta * pA[c];
                    pT[c] = wT[c] + beta * pT[c];
                }
            }
            orc_Amul(s, wA, pA);
            orc_Tmul(s, wT, pT);
            const double wApT = orc_gSumProd(s, wA, pT);
            if (check_singularity(&perf, fabs(wApT) / normFactor)) break;
            const double alpha = wArT / wApT;
            for (int c = 0; c < n; c++)
            {
                psi[c] += alpha * pA[c];
                rA[c] -= alpha * wA[c];
                rT[c] -= alpha * wT[c];
            }
            perf.finalResidual = orc_gSumMag(s, rA) / normFactor;
            hist_push(&perf, hist);
        } while (perf.nIterations++ < o->maxIter
                 && !check_convergence(&perf, o->tolerance, o->relTol));
        precond_free(s, P);
    }
    free(pA); free(pT); free(wA); free(wT); free(rA); free(rT);
    return perf;
}

/* ------------------------------------------------------------------ smoothSolver
 * solvers/smoothSolver/smoothSolver.C:77-180 */
static orc_perf solve_smooth(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                             double* hist)
{
    orc_perf perf; memset(&perf, 0, sizeof(perf));
    const int n = s->nCellsTotal;
    if (o->nSweeps < 0)
    {
        orc_smooth(s, o->smoother, psi, source, -o->nSweeps);
        perf.nIterations -= o->nSweeps;
        return perf;
    }
    double* Apsi = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* temp = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    orc_Amul(s, Apsi, psi);
    const double normFactor = orc_normFactor(s, psi, source, Apsi, temp);
    perf.normFactor = normFactor;
    for (int i = 0; i < n; i++) temp[i] = source[i] - Apsi[i];
    perf.initialResidual = orc_gSumMag(s, temp) / normFactor;
    perf.finalResidual = perf.initialResidual;
    hist_push(&perf, hist);
    if (!check_convergence(&perf, o->tolerance, o->relTol))
    {
        do
        {
            orc_smooth(s, o->smoother, psi, source, o->nSweeps);
            orc_residual(s, temp, psi, source);
            perf.finalResidual = orc_gSumMag(s, temp) / normFactor;
            hist_push(&perf, hist);
        } while ((perf.nIterations += o->nSweeps) < o->maxIter
                 && !check_convergence(&perf, o->tolerance, o->relTol));
    }
    free(Apsi); free(temp);
    return perf;
}

/* solvers/diagonalSolver/diagonalSolver.C:62-81 */
static orc_perf solve_diagonal(const orc_sys* s, double* psi, const double* source)
{
    orc_perf perf; memset(&perf, 0, sizeof(perf));
    for (int d = 0; d < s->nDom; d++)
    {
        const orc_dom* D = &s->dom[d];
        for (int c = 0; c < D->nCells; c++)
            psi[D->cellOffset + c] = source[D->cellOffset + c] / D->diag[c];
    }
    perf.converged = 1; /* solverPerformance(typeName, fieldName, 0, 0, 0, true, false) */
    return perf;
}

orc_perf orc_solve(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                   const double* faceWeights, double* resHist)
{
    switch (o->solver)
    {
    case ORC_PCG:      return solve_pcg(s, o, psi, source, faceWeights, resHist);
    case ORC_PBICG:    return solve_pbicg(s, o, psi, source, resHist);
    case ORC_SMOOTH:   return solve_smooth(s, o, psi, source, resHist);
    case ORC_GAMG:     return orc_gamg_solve(s, o, psi, source, faceWeights, resHist);
    case ORC_DIAGONAL: return solve_diagonal(s, psi, source);
    }
    orc_perf p; memset(&p, 0, sizeof(p));
    return p;
}
