/*
 * TEST INFRASTRUCTURE - CPU oracle, part 2: GAMG (pair agglomeration, level
 * matrices, V-cycle, GAMG preconditioner).  See ldu_oracle.h.
 * Reference paths relative to
 * /root/reference/src/OpenFOAM/matrices/lduMatrix/solvers/GAMG/ unless stated.
 */
#include "ldu_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_GREAT 1e20
#define ORC_SMALL 1e-20
#define ORC_VSMALL 1e-300
#define ORC_MAXLEVELS 50 /* GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomeration.C:75 */

/* Coarse image of one processor patch (processorGAMGInterface.C:47-126). */
typedef struct lvl_patch {
    int n;          /* coarse patch faces */
    int nFine;      /* faces of the fine patch it was built from */
    int* faceCells; /* [n] coarse cell of each coarse patch face */
    int* fra;       /* [nFine] faceRestrictAddressing: fine patch face -> coarse patch face */
    double* bou;    /* [n] agglomerated interfaceBouCoeffs */
    double* intc;   /* [n] agglomerated interfaceIntCoeffs */
} lvl_patch;

/* One coarse level: owns its addressing, maps and coefficients, per domain. */
typedef struct dom_level {
    int nFineCells, nFineFaces;
    int nCells, nFaces;
    int* restrictAddr;     /* [nFineCells] fine cell -> coarse cell */
    int* faceRestrictAddr; /* [nFineFaces] fine face -> coarse face, or -(coarse cell)-1 */
    int* l;                /* coarse lowerAddr */
    int* u;                /* coarse upperAddr */
    double* diag;
    double* upper;
    double* lower;         /* NULL when symmetric */
    orc_patch* patches;    /* coarse coupled patches (owned) */
    int nPatches;
    lvl_patch* lp;         /* [nPatches] */
} dom_level;

typedef struct level_t {
    dom_level* dl;         /* [nDom] */
    orc_sys sys;           /* the coarse system (Amul, smoothers, coarsest solve) */
} level_t;

struct orc_gamg {
    const orc_sys* fine;
    int nDom;
    int nLevels;
    int sym;
    level_t lev[ORC_MAXLEVELS];
};

/* ------------------------------------------------------------------
 * GAMGAgglomerations/pairGAMGAgglomeration/pairGAMGAgglomerate.C:31-198 */
static int* pair_agglomerate(int* nCoarseCellsOut, int nFineCells, int nFaces,
                             const int* lowerAddr, const int* upperAddr,
                             const double* faceWeights)
{
    int* cellFaces = (int*)malloc(sizeof(int) * (size_t)(2 * nFaces + 1));
    int* cellFaceOffsets = (int*)malloc(sizeof(int) * (size_t)(nFineCells + 1));
    int* nNbrs = (int*)calloc((size_t)nFineCells + 1, sizeof(int));

    for (int f = 0; f < nFaces; f++) nNbrs[upperAddr[f]]++;
    for (int f = 0; f < nFaces; f++) nNbrs[lowerAddr[f]]++;
    cellFaceOffsets[0] = 0;
    for (int c = 0; c < nFineCells; c++) cellFaceOffsets[c + 1] = cellFaceOffsets[c] + nNbrs[c];
    for (int c = 0; c < nFineCells; c++) nNbrs[c] = 0;
    for (int f = 0; f < nFaces; f++)
    {
        cellFaces[cellFaceOffsets[upperAddr[f]] + nNbrs[upperAddr[f]]] = f;
        nNbrs[upperAddr[f]]++;
    }
    for (int f = 0; f < nFaces; f++)
    {
        cellFaces[cellFaceOffsets[lowerAddr[f]] + nNbrs[lowerAddr[f]]] = f;
        nNbrs[lowerAddr[f]]++;
    }

    int* coarseCellMap = (int*)malloc(sizeof(int) * (size_t)(nFineCells + 1));
    for (int c = 0; c < nFineCells; c++) coarseCellMap[c] = -1;
    int nCoarseCells = 0;

    for (int celli = 0; celli < nFineCells; celli++)
    {
        if (coarseCellMap[celli] < 0)
        {
            int matchFaceNo = -1;
            double maxFaceWeight = -ORC_GREAT;
            for (int faceOs = cellFaceOffsets[celli]; faceOs < cellFaceOffsets[celli + 1]; faceOs++)
            {
                int facei = cellFaces[faceOs];
                if (coarseCellMap[upperAddr[facei]] < 0
                    && coarseCellMap[lowerAddr[facei]] < 0
                    && faceWeights[facei] > maxFaceWeight)
                {
                    matchFaceNo = facei;
                    maxFaceWeight = faceWeights[facei];
                }
            }
            if (matchFaceNo >= 0)
            {
                coarseCellMap[upperAddr[matchFaceNo]] = nCoarseCells;
                coarseCellMap[lowerAddr[matchFaceNo]] = nCoarseCells;
                nCoarseCells++;
            }
            else
            {
                int clusterMatchFaceNo = -1;
                double clusterMaxFaceCoeff = -ORC_GREAT;
                for (int faceOs = cellFaceOffsets[celli]; faceOs < cellFaceOffsets[celli + 1]; faceOs++)
                {
                    int facei = cellFaces[faceOs];
                    if (faceWeights[facei] > clusterMaxFaceCoeff)
                    {
                        clusterMatchFaceNo = facei;
                        clusterMaxFaceCoeff = faceWeights[facei];
                    }
                }
                if (clusterMatchFaceNo >= 0)
                {
                    int a = coarseCellMap[upperAddr[clusterMatchFaceNo]];
                    int b = coarseCellMap[lowerAddr[clusterMatchFaceNo]];
                    coarseCellMap[celli] = a > b ? a : b;
                }
            }
        }
    }
    for (int celli = 0; celli < nFineCells; celli++)
    {
        if (coarseCellMap[celli] < 0)
        {
            coarseCellMap[celli] = nCoarseCells;
            nCoarseCells++;
        }
    }
    /* reverse the map ordering (:186-195) */
    nCoarseCells--;
    for (int celli = 0; celli < nFineCells; celli++)
        coarseCellMap[celli] = nCoarseCells - coarseCellMap[celli];
    nCoarseCells++;

    free(cellFaces); free(cellFaceOffsets); free(nNbrs);
    *nCoarseCellsOut = nCoarseCells;
    return coarseCellMap;
}

/* ------------------------------------------------------------------
 * GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomerateLduAddressing.C:31-198 */
static void agglomerate_addressing(dom_level* L, const int* lowerAddr, const int* upperAddr)
{
    const int nFineFaces = L->nFineFaces;
    const int nCoarseCells = L->nCells;
    const int* restrictMap = L->restrictAddr;

    int maxNnbrs = 10;
    int* cCellnFaces = (int*)calloc((size_t)nCoarseCells + 1, sizeof(int));
    int* cCellFaces = (int*)malloc(sizeof(int) * (size_t)maxNnbrs * (size_t)(nCoarseCells + 1));
    int* faceRestrictAddr = (int*)malloc(sizeof(int) * (size_t)(nFineFaces + 1));
    int* initCoarseNeighb = (int*)malloc(sizeof(int) * (size_t)(nFineFaces + 1));
    int nCoarseFaces = 0;

    for (int fineFacei = 0; fineFacei < nFineFaces; fineFacei++)
    {
        int rmUpperAddr = restrictMap[upperAddr[fineFacei]];
        int rmLowerAddr = restrictMap[lowerAddr[fineFacei]];
        if (rmUpperAddr == rmLowerAddr)
        {
            faceRestrictAddr[fineFacei] = -(rmUpperAddr + 1);
        }
        else
        {
            int cOwn = rmUpperAddr;
            int cNei = rmLowerAddr;
            if (rmUpperAddr > rmLowerAddr)
            {
                cOwn = rmLowerAddr;
                cNei = rmUpperAddr;
            }
            int* ccFaces = &cCellFaces[(size_t)maxNnbrs * cOwn];
            int nbrFound = 0;
            int* ccnFaces = &cCellnFaces[cOwn];
            for (int i = 0; i < *ccnFaces; i++)
            {
                if (initCoarseNeighb[ccFaces[i]] == cNei)
                {
                    nbrFound = 1;
                    faceRestrictAddr[fineFacei] = ccFaces[i];
                    break;
                }
            }
            if (!nbrFound)
            {
                if (*ccnFaces >= maxNnbrs)
                {
                    int oldMaxNnbrs = maxNnbrs;
                    maxNnbrs *= 2;
                    cCellFaces = (int*)realloc(cCellFaces,
                        sizeof(int) * (size_t)maxNnbrs * (size_t)(nCoarseCells + 1));
                    for (int i = nCoarseCells - 1; i >= 0; i--)
                    {
                        int* oldCcNbrs = &cCellFaces[(size_t)oldMaxNnbrs * i];
                        int* newCcNbrs = &cCellFaces[(size_t)maxNnbrs * i];
                        for (int j = cCellnFaces[i] - 1; j >= 0; j--) newCcNbrs[j] = oldCcNbrs[j];
                    }
                    ccFaces = &cCellFaces[(size_t)maxNnbrs * cOwn];
                }
                ccFaces[*ccnFaces] = nCoarseFaces;
                initCoarseNeighb[nCoarseFaces] = cNei;
                faceRestrictAddr[fineFacei] = nCoarseFaces;
                (*ccnFaces)++;
                nCoarseFaces++;
            }
        }
    }

    /* renumber into upper-triangular order (:167-198) */
    int* coarseOwner = (int*)malloc(sizeof(int) * (size_t)(nCoarseFaces + 1));
    int* coarseNeighbour = (int*)malloc(sizeof(int) * (size_t)(nCoarseFaces + 1));
    int* coarseFaceMap = (int*)malloc(sizeof(int) * (size_t)(nCoarseFaces + 1));
    int coarseFacei = 0;
    for (int cci = 0; cci < nCoarseCells; cci++)
    {
        int* cFaces = &cCellFaces[(size_t)maxNnbrs * cci];
        int ccnFaces = cCellnFaces[cci];
        for (int i = 0; i < ccnFaces; i++)
        {
            coarseOwner[coarseFacei] = cci;
            coarseNeighbour[coarseFacei] = initCoarseNeighb[cFaces[i]];
            coarseFaceMap[cFaces[i]] = coarseFacei;
            coarseFacei++;
        }
    }
    for (int fineFacei = 0; fineFacei < nFineFaces; fineFacei++)
        if (faceRestrictAddr[fineFacei] >= 0)
            faceRestrictAddr[fineFacei] = coarseFaceMap[faceRestrictAddr[fineFacei]];

    free(cCellnFaces); free(cCellFaces); free(initCoarseNeighb); free(coarseFaceMap);
    L->faceRestrictAddr = faceRestrictAddr;
    L->nFaces = nCoarseFaces;
    L->l = coarseOwner;
    L->u = coarseNeighbour;
}

static void free_dom_level_addr(dom_level* L)
{
    for (int p = 0; p < L->nPatches; p++)
    {
        free(L->lp[p].faceCells); free(L->lp[p].fra); free(L->lp[p].bou); free(L->lp[p].intc);
    }
    free(L->lp);
    free(L->restrictAddr); free(L->faceRestrictAddr); free(L->l); free(L->u);
    free(L->diag); free(L->upper); free(L->lower); free(L->patches);
    memset(L, 0, sizeof(*L));
}

/* pairGAMGAgglomerationCombineLevels.C:32-95 : fold level `cur` into `prev` */
static void combine_levels(dom_level* prev, dom_level* cur)
{
    for (int i = 0; i < prev->nFineFaces; i++)
    {
        if (prev->faceRestrictAddr[i] >= 0)
            prev->faceRestrictAddr[i] = cur->faceRestrictAddr[prev->faceRestrictAddr[i]];
        else
            prev->faceRestrictAddr[i] = -cur->restrictAddr[-prev->faceRestrictAddr[i] - 1] - 1;
    }
    for (int i = 0; i < prev->nFineCells; i++)
        prev->restrictAddr[i] = cur->restrictAddr[prev->restrictAddr[i]];
    /* GAMGInterface::combine (GAMGInterface.C:36-49) */
    for (int p = 0; p < prev->nPatches; p++)
    {
        lvl_patch* P = &prev->lp[p];
        lvl_patch* C = &cur->lp[p];
        for (int i = 0; i < P->nFine; i++) P->fra[i] = C->fra[P->fra[i]];
        free(P->faceCells);
        P->faceCells = C->faceCells; C->faceCells = 0;
        P->n = C->n;
    }
    prev->nCells = cur->nCells;
    prev->nFaces = cur->nFaces;
    free(prev->l); free(prev->u);
    prev->l = cur->l; prev->u = cur->u;
    cur->l = cur->u = 0;
    free_dom_level_addr(cur);
}

/* ------------------------------------------------------------------
 * GAMGSolverAgglomerateMatrix.C:31-207 (one domain; interfaces handled by the caller) */
static void agglomerate_matrix(dom_level* L, int nFineCells, int nFineFaces, const int* fineL,
                               const double* fineDiag, const double* fineUpper,
                               const double* fineLower /* NULL if symmetric */)
{
    L->diag = (double*)calloc((size_t)L->nCells + 1, sizeof(double));
    L->upper = (double*)calloc((size_t)L->nFaces + 1, sizeof(double));
    L->lower = fineLower ? (double*)calloc((size_t)L->nFaces + 1, sizeof(double)) : 0;

    /* restrictField (GAMGAgglomerationTemplates.C:31-59) */
    for (int i = 0; i < nFineCells; i++) L->diag[L->restrictAddr[i]] += fineDiag[i];

    if (fineLower)
    {
        for (int fineFacei = 0; fineFacei < nFineFaces; fineFacei++)
        {
            int cFace = L->faceRestrictAddr[fineFacei];
            if (cFace >= 0)
            {
                if (L->l[cFace] == L->restrictAddr[fineL[fineFacei]])
                {
                    L->upper[cFace] += fineUpper[fineFacei];
                    L->lower[cFace] += fineLower[fineFacei];
                }
                else if (L->u[cFace] == L->restrictAddr[fineL[fineFacei]])
                {
                    L->upper[cFace] += fineLower[fineFacei];
                    L->lower[cFace] += fineUpper[fineFacei];
                }
                else
                {
                    fprintf(stderr, "orc: inconsistent addressing between fine and coarse grids\n");
                    abort();
                }
            }
            else
            {
                L->diag[-1 - cFace] += fineUpper[fineFacei] + fineLower[fineFacei];
            }
        }
    }
    else
    {
        for (int fineFacei = 0; fineFacei < nFineFaces; fineFacei++)
        {
            int cFace = L->faceRestrictAddr[fineFacei];
            if (cFace >= 0) L->upper[cFace] += fineUpper[fineFacei];
            else L->diag[-1 - cFace] += 2 * fineUpper[fineFacei];
        }
    }
}

/* ------------------------------------------------------------------
 * build: GAMGSolver.C:44-127 + pairGAMGAgglomerate.C:201-292 */
orc_gamg* orc_gamg_build(const orc_sys* s, const orc_opts* o, const double* faceWeightsAll)
{
    orc_gamg* g = (orc_gamg*)calloc(1, sizeof(orc_gamg));
    g->fine = s;
    g->nDom = s->nDom;
    g->sym = 1;
    for (int d = 0; d < s->nDom; d++)
    {
        if (s->dom[d].lower != s->dom[d].upper) g->sym = 0;
    }
    const int nDom = s->nDom;

    /* per-domain face weights of the current finest pair level */
    double** w = (double**)calloc((size_t)nDom, sizeof(double*));
    {
        int fo = 0;
        for (int d = 0; d < nDom; d++)
        {
            const orc_dom* D = &s->dom[d];
            w[d] = (double*)malloc(sizeof(double) * (size_t)(D->nFaces + 1));
            for (int f = 0; f < D->nFaces; f++)
            {
                if (o->agglomerator == ORC_AGG_ALGEBRAICPAIR)
                    /* algebraicPairGAMGAgglomeration.C:55: mag(matrix.upper()) */
                    w[d][f] = fabs(D->upper[f]);
                else
                    w[d][f] = faceWeightsAll[fo + f];
            }
            fo += D->nFaces;
        }
    }

    int nPairLevels = 0;
    int nCreatedLevels = 0;
    while (nCreatedLevels < ORC_MAXLEVELS - 1)
    {
        level_t* Lv = &g->lev[nCreatedLevels];
        Lv->dl = (dom_level*)calloc((size_t)nDom, sizeof(dom_level));
        int contAgg = 1;
        for (int d = 0; d < nDom; d++)
        {
            dom_level* L = &Lv->dl[d];
            int nFC, nFF; const int *fl, *fu;
            if (nCreatedLevels == 0)
            {
                nFC = s->dom[d].nCells; nFF = s->dom[d].nFaces; fl = s->dom[d].l; fu = s->dom[d].u;
            }
            else
            {
                dom_level* P = &g->lev[nCreatedLevels - 1].dl[d];
                nFC = P->nCells; nFF = P->nFaces; fl = P->l; fu = P->u;
            }
            L->nFineCells = nFC; L->nFineFaces = nFF;
            L->restrictAddr = pair_agglomerate(&L->nCells, nFC, nFF, fl, fu, w[d]);
            /* continueAgglomerating: GAMGAgglomeration.C:53-62 (and-reduce over ranks) */
            if (!(L->nCells >= o->nCellsInCoarsestLevel)) contAgg = 0;
        }
        if (!contAgg)
        {
            for (int d = 0; d < nDom; d++) free_dom_level_addr(&Lv->dl[d]);
            free(Lv->dl); Lv->dl = 0;
            break;
        }
        for (int d = 0; d < nDom; d++)
        {
            dom_level* L = &Lv->dl[d];
            const int *fl, *fu;
            if (nCreatedLevels == 0) { fl = s->dom[d].l; fu = s->dom[d].u; }
            else { fl = g->lev[nCreatedLevels - 1].dl[d].l; fu = g->lev[nCreatedLevels - 1].dl[d].u; }
            agglomerate_addressing(L, fl, fu);
            /* restrictFaceField (GAMGAgglomerationTemplates.C:63-83) */
            double* aw = (double*)calloc((size_t)L->nFaces + 1, sizeof(double));
            for (int ff = 0; ff < L->nFineFaces; ff++)
            {
                int cFace = L->faceRestrictAddr[ff];
                if (cFace >= 0) aw[cFace] += w[d][ff];
            }
            free(w[d]);
            w[d] = aw;
        }
        /* coarse processor interfaces (GAMGAgglomerateLduAddressing.C:201-268 +
         * processorGAMGInterface.C:47-126): coarse patch faces = unique (master cell, slave cell)
         * pairs in order of first occurrence, so both ranks number them identically */
        for (int d = 0; d < nDom; d++)
        {
            dom_level* L = &Lv->dl[d];
            const int nP = s->dom[d].nPatches;
            L->nPatches = nP;
            L->lp = (lvl_patch*)calloc((size_t)nP + 1, sizeof(lvl_patch));
            for (int p = 0; p < nP; p++)
            {
                const int q = s->dom[d].patches[p].nbrDom;
                const int pq = s->dom[d].patches[p].nbrPatch;
                const int* Fd; const int* Fq; int nFine;
                if (nCreatedLevels == 0)
                {
                    Fd = s->dom[d].patches[p].faceCells; Fq = s->dom[q].patches[pq].faceCells;
                    nFine = s->dom[d].patches[p].n;
                }
                else
                {
                    Fd = g->lev[nCreatedLevels - 1].dl[d].lp[p].faceCells;
                    Fq = g->lev[nCreatedLevels - 1].dl[q].lp[pq].faceCells;
                    nFine = g->lev[nCreatedLevels - 1].dl[d].lp[p].n;
                }
                const int* rd = L->restrictAddr;
                const int* rq = Lv->dl[q].restrictAddr;
                lvl_patch* P = &L->lp[p];
                P->nFine = nFine;
                P->fra = (int*)malloc(sizeof(int) * (size_t)(nFine + 1));
                P->faceCells = (int*)malloc(sizeof(int) * (size_t)(nFine + 1));
                int* pairA = (int*)malloc(sizeof(int) * (size_t)(nFine + 1));
                int* pairB = (int*)malloc(sizeof(int) * (size_t)(nFine + 1));
                int nc = 0;
                for (int ffi = 0; ffi < nFine; ffi++)
                {
                    const int loc = rd[Fd[ffi]], nbr = rq[Fq[ffi]];
                    const int a = d < q ? loc : nbr;   /* master side first */
                    const int b = d < q ? nbr : loc;
                    int found = -1;
                    for (int c = 0; c < nc; c++) if (pairA[c] == a && pairB[c] == b) { found = c; break; }
                    if (found < 0)
                    {
                        pairA[nc] = a; pairB[nc] = b;
                        P->faceCells[nc] = loc;
                        found = nc++;
                    }
                    P->fra[ffi] = found;
                }
                P->n = nc;
                free(pairA); free(pairB);
            }
        }
        if (nPairLevels % o->mergeLevels)
        {
            for (int d = 0; d < nDom; d++)
                combine_levels(&g->lev[nCreatedLevels - 1].dl[d], &Lv->dl[d]);
            free(Lv->dl); Lv->dl = 0;
        }
        else
        {
            nCreatedLevels++;
        }
        nPairLevels++;
    }
    for (int d = 0; d < nDom; d++) free(w[d]);
    free(w);
    g->nLevels = nCreatedLevels;

    /* level matrices: GAMGSolver.C:86-89 */
    for (int lev = 0; lev < g->nLevels; lev++)
    {
        level_t* Lv = &g->lev[lev];
        Lv->sys.nDom = nDom;
        Lv->sys.dom = (orc_dom*)calloc((size_t)nDom, sizeof(orc_dom));
        for (int d = 0; d < nDom; d++)
        {
            dom_level* L = &Lv->dl[d];
            const orc_dom* F = (lev == 0) ? &s->dom[d] : &g->lev[lev - 1].sys.dom[d];
            const double* fLower = (F->lower != F->upper) ? F->lower : 0;
            agglomerate_matrix(L, F->nCells, F->nFaces, F->l, F->diag, F->upper, fLower);
            orc_dom* C = &Lv->sys.dom[d];
            C->nCells = L->nCells; C->nFaces = L->nFaces;
            C->l = L->l; C->u = L->u;
            C->diag = L->diag; C->upper = L->upper;
            C->lower = L->lower ? L->lower : L->upper;
            /* GAMGInterface::agglomerateCoeffs (GAMGInterface.C:61-75) */
            C->nPatches = L->nPatches;
            L->patches = (orc_patch*)calloc((size_t)L->nPatches + 1, sizeof(orc_patch));
            for (int p = 0; p < L->nPatches; p++)
            {
                lvl_patch* P = &L->lp[p];
                const orc_patch* FP = &F->patches[p];
                P->bou = (double*)calloc((size_t)P->n + 1, sizeof(double));
                P->intc = (double*)calloc((size_t)P->n + 1, sizeof(double));
                for (int ffi = 0; ffi < P->nFine; ffi++)
                {
                    P->bou[P->fra[ffi]] += FP->bouCoeffs[ffi];
                    P->intc[P->fra[ffi]] += FP->intCoeffs[ffi];
                }
                L->patches[p].n = P->n;
                L->patches[p].faceCells = P->faceCells;
                L->patches[p].bouCoeffs = P->bou;
                L->patches[p].intCoeffs = P->intc;
                L->patches[p].nbrDom = FP->nbrDom;
                L->patches[p].nbrPatch = FP->nbrPatch;
            }
            C->patches = L->patches;
        }
        orc_sys_finalize(&Lv->sys);
    }
    return g;
}

void orc_gamg_free(orc_gamg* g)
{
    if (!g) return;
    for (int lev = 0; lev < g->nLevels; lev++)
    {
        level_t* Lv = &g->lev[lev];
        orc_sys_free_derived(&Lv->sys);
        for (int d = 0; d < g->nDom; d++) free_dom_level_addr(&Lv->dl[d]);
        free(Lv->dl);
        free(Lv->sys.dom);
    }
    free(g);
}

int orc_gamg_nLevels(const orc_gamg* g) { return g->nLevels; }
int orc_gamg_level_nCells(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].nCells; }
int orc_gamg_level_nFaces(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].nFaces; }
/* multi-domain hierarchies: cells / faces of level lev on domain d (the levels are common to all domains: the and-reduce of
   continueAgglomerating, GAMGAgglomeration.C:53-62) */
int orc_gamg_level_nCells_dom(const orc_gamg* g, int lev, int d) { return g->lev[lev].dl[d].nCells; }
int orc_gamg_level_nFaces_dom(const orc_gamg* g, int lev, int d) { return g->lev[lev].dl[d].nFaces; }
const int* orc_gamg_restrict(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].restrictAddr; }
const int* orc_gamg_faceRestrict(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].faceRestrictAddr; }
const int* orc_gamg_level_lower(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].l; }
const int* orc_gamg_level_upper(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].u; }
const double* orc_gamg_level_diag(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].diag; }
const double* orc_gamg_level_upperCoeffs(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].upper; }
const double* orc_gamg_level_lowerCoeffs(const orc_gamg* g, int lev) { return g->lev[lev].dl[0].lower; }

/* ------------------------------------------------------------------ V-cycle helpers */

static const orc_sys* level_sys(const orc_gamg* g, int i /* 0 = finest */)
{
    return i == 0 ? g->fine : &g->lev[i - 1].sys;
}

/* GAMGAgglomerationTemplates.C:31-59: cf = 0; cf[map[i]] += ff[i]  (fineLevelIndex) */
static void restrict_field(const orc_gamg* g, double* cf, const double* ff, int fineLevelIndex)
{
    const orc_sys* cs = &g->lev[fineLevelIndex].sys;
    const orc_sys* fs = level_sys(g, fineLevelIndex);
    ORC_PAR
    for (int d = 0; d < g->nDom; d++)
    {
        const dom_level* L = &g->lev[fineLevelIndex].dl[d];
        double* c = cf + cs->dom[d].cellOffset;
        const double* f = ff + fs->dom[d].cellOffset;
        for (int i = 0; i < L->nCells; i++) c[i] = 0;
        for (int i = 0; i < L->nFineCells; i++) c[L->restrictAddr[i]] += f[i];
    }
}

/* GAMGAgglomerationTemplates.C:87-100: ff[i] = cf[map[i]]  (coarseLevelIndex) */
static void prolong_field(const orc_gamg* g, double* ff, const double* cf, int coarseLevelIndex)
{
    const orc_sys* cs = &g->lev[coarseLevelIndex].sys;
    const orc_sys* fs = level_sys(g, coarseLevelIndex);
    ORC_PAR
    for (int d = 0; d < g->nDom; d++)
    {
        const dom_level* L = &g->lev[coarseLevelIndex].dl[d];
        const double* c = cf + cs->dom[d].cellOffset;
        double* f = ff + fs->dom[d].cellOffset;
        for (int i = 0; i < L->nFineCells; i++) f[i] = c[L->restrictAddr[i]];
    }
}

/* GAMGSolverScale.C:31-75 */
static void gamg_scale(const orc_sys* A, double* field, double* Acf, const double* source)
{
    orc_Amul(A, Acf, field);
    double num = 0, den = 0;
    double pn[ORC_MAXDOM], pd[ORC_MAXDOM];
    ORC_PAR
    for (int d = 0; d < A->nDom; d++)
    {
        const orc_dom* D = &A->dom[d];
        double scalingFactorNum = 0.0, scalingFactorDenom = 0.0;
        for (int i = D->cellOffset; i < D->cellOffset + D->nCells; i++)
        {
            scalingFactorNum += source[i] * field[i];
            scalingFactorDenom += Acf[i] * field[i];
        }
        pn[d] = scalingFactorNum;
        pd[d] = scalingFactorDenom;
    }
    for (int d = 0; d < A->nDom; d++)
    {
        num = d == 0 ? pn[d] : num + pn[d];
        den = d == 0 ? pd[d] : den + pd[d];
    }
    /* stabilise(y, VSMALL): src/OpenFOAM/primitives/Scalar/Scalar.H: y<0 ? y-small : y+small */
    const double stab = den < 0 ? den - ORC_VSMALL : den + ORC_VSMALL;
    const double sf = num / stab;
    ORC_PAR
    for (int d = 0; d < A->nDom; d++)
    {
        const orc_dom* D = &A->dom[d];
        for (int i = 0; i < D->nCells; i++)
        {
            const int k = D->cellOffset + i;
            field[k] = sf * field[k] + (source[k] - sf * Acf[k]) / D->diag[i];
        }
    }
}

/* GAMGSolverInterpolate.C:30-83 (coupled patches: same update as Amul) */
static void gamg_interpolate(const orc_sys* A, double* psi, double* Apsi)
{
    /* Apsi = (L+U) psi - coupled, computed as Amul(psi) - D psi would change rounding, so
     * restate the loops directly. */
    for (int d = 0; d < A->nDom; d++)
    {
        const orc_dom* D = &A->dom[d];
        double* y = Apsi + D->cellOffset;
        const double* x = psi + D->cellOffset;
        for (int c = 0; c < D->nCells; c++) y[c] = 0;
        for (int face = 0; face < D->nFaces; face++)
        {
            y[D->u[face]] += D->lower[face] * x[D->l[face]];
            y[D->l[face]] += D->upper[face] * x[D->u[face]];
        }
        for (int p = 0; p < D->nPatches; p++)
        {
            const orc_patch* P = &D->patches[p];
            const orc_dom* N = &A->dom[P->nbrDom];
            const orc_patch* NP = &N->patches[P->nbrPatch];
            for (int i = 0; i < P->n; i++)
                y[P->faceCells[i]] -= P->bouCoeffs[i] * psi[N->cellOffset + NP->faceCells[i]];
        }
    }
    for (int d = 0; d < A->nDom; d++)
    {
        const orc_dom* D = &A->dom[d];
        for (int c = 0; c < D->nCells; c++)
            psi[D->cellOffset + c] = -Apsi[D->cellOffset + c] / D->diag[c];
    }
}

typedef struct vcycle_ws {
    double** coarseCorr;   /* [nLevels] */
    double** coarseSrc;    /* [nLevels] */
} vcycle_ws;

static vcycle_ws ws_new(const orc_gamg* g)
{
    vcycle_ws w;
    w.coarseCorr = (double**)calloc((size_t)g->nLevels, sizeof(double*));
    w.coarseSrc = (double**)calloc((size_t)g->nLevels, sizeof(double*));
    for (int i = 0; i < g->nLevels; i++)
    {
        int n = g->lev[i].sys.nCellsTotal;
        w.coarseCorr[i] = (double*)calloc((size_t)n + 1, sizeof(double));
        w.coarseSrc[i] = (double*)calloc((size_t)n + 1, sizeof(double));
    }
    return w;
}
static void ws_free(const orc_gamg* g, vcycle_ws* w)
{
    for (int i = 0; i < g->nLevels; i++) { free(w->coarseCorr[i]); free(w->coarseSrc[i]); }
    free(w->coarseCorr); free(w->coarseSrc);
}

static int imin(int a, int b) { return a < b ? a : b; }

/* directSolveCoarsest (GAMGSolver.C:95-106, GAMGSolverSolve.C:436-440): the coarsest level as a dense matrix, Crout LU with
 * implicit scaled partial pivoting (scalarMatrices.C:31-134 LUDecompose) and LUBacksubstitute (scalarMatricesTemplates.C:119-164)
 * - every loop in the reference's order.
 * One domain (serial run; LUscalarMatrix.C:128-187 convert): diag, lower/upper by face, cyclic interfaces:
 *   A[faceCells][nbr faceCells] -= the neighbour patch's interfaceBouCoeffs.
 * Several domains = the ranks of a parallel run (LUscalarMatrix.C:52-107): every rank's matrix travels to the master, which
 * builds ONE dense matrix over all ranks' cells (rank r's cells at procOffsets[r] = the concatenation used everywhere in this
 * oracle) - :190-318 convert(lduMatrices): per rank diag, faces; per processor interface with myProcNo < neighbProcNo BOTH
 * coupling entries ( A[u][l] -= the NEIGHBOUR interface's coeffs, A[l][u] -= this interface's coeffs ) - factorises it there,
 * and per solve gathers the sources, back-substitutes and scatters (LUscalarMatrixTemplates.C:31-118).  A cyclic patch inside a
 * rank of a parallel run (myProcNo == neighbProcNo == -1 in procLduInterface.C:38-60) is read by :247-262 as ONE patch holding both
 * halves - the pre-2.0 layout, which no 2.2.x mesh has: not restated, refused. */
static void lu_direct_solve(const orc_sys* A, double* x, const double* src)
{
    const int n = A->nCellsTotal;
    double* M = (double*)calloc((size_t)n * n + 1, sizeof(double));
    int* piv = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    double* vv = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
#define MM(i, j) M[(size_t)(i) * n + (j)]
    for (int d = 0; d < A->nDom; d++)
    {
        const orc_dom* D = &A->dom[d];
        const int off = D->cellOffset;
        for (int c = 0; c < D->nCells; c++) MM(off + c, off + c) = D->diag[c];
        for (int f = 0; f < D->nFaces; f++)
        {
            MM(off + D->u[f], off + D->l[f]) = D->lower[f];
            MM(off + D->l[f], off + D->u[f]) = D->upper[f];
        }
        for (int p = 0; p < D->nPatches; p++)
        {
            const orc_patch* P = &D->patches[p];
            if (A->nDom == 1)
            {
                const orc_patch* N = &D->patches[P->nbrPatch];      /* serial: coupled patches are cyclic pairs */
                for (int f = 0; f < P->n; f++) MM(P->faceCells[f], N->faceCells[f]) -= N->bouCoeffs[f];
                continue;
            }
            if (P->nbrDom == d)
            {
                fprintf(stderr, "oracle: directSolveCoarsest: a cyclic patch inside a rank of a parallel run is not restated\n");
                abort();
            }
            if (d > P->nbrDom) continue;                             /* :264 myProcNo_ < neighbProcNo_ */
            const orc_dom* DN = &A->dom[P->nbrDom];
            const orc_patch* N = &DN->patches[P->nbrPatch];
            const int noff = DN->cellOffset;
            for (int f = 0; f < P->n; f++)                            /* :308-315 */
            {
                const int uCell = P->faceCells[f] + off, lCell = N->faceCells[f] + noff;
                MM(uCell, lCell) -= N->bouCoeffs[f];
                MM(lCell, uCell) -= P->bouCoeffs[f];
            }
        }
    }
    /* LUDecompose */
    for (int i = 0; i < n; i++)
    {
        double largest = 0.0, t;
        for (int j = 0; j < n; j++) if ((t = fabs(MM(i, j))) > largest) largest = t;
        if (largest == 0.0) { fprintf(stderr, "oracle: LUdecompose: Singular matrix\n"); abort(); }
        vv[i] = 1.0 / largest;
    }
    for (int j = 0; j < n; j++)
    {
        for (int i = 0; i < j; i++)
        {
            double sum = MM(i, j);
            for (int k = 0; k < i; k++) sum -= MM(i, k) * MM(k, j);
            MM(i, j) = sum;
        }
        int iMax = 0;
        double largest = 0.0;
        for (int i = j; i < n; i++)
        {
            double sum = MM(i, j);
            for (int k = 0; k < j; k++) sum -= MM(i, k) * MM(k, j);
            MM(i, j) = sum;
            const double t = vv[i] * fabs(sum);
            if (t >= largest) { largest = t; iMax = i; }
        }
        piv[j] = iMax;
        if (j != iMax)
        {
            for (int k = 0; k < n; k++) { const double t = MM(j, k); MM(j, k) = MM(iMax, k); MM(iMax, k) = t; }
            vv[iMax] = vv[j];
        }
        if (MM(j, j) == 0.0) MM(j, j) = 1e-15;   /* SMALL */
        if (j != n - 1)
        {
            const double rDiag = 1.0 / MM(j, j);
            for (int i = j + 1; i < n; i++) MM(i, j) *= rDiag;
        }
    }
    /* coarsestCorrField = coarsestSource; LUBacksubstitute */
    for (int i = 0; i < n; i++) x[i] = src[i];
    int ii = 0;
    for (int i = 0; i < n; i++)
    {
        const int ip = piv[i];
        double sum = x[ip];
        x[ip] = x[i];
        if (ii != 0) { for (int j = ii - 1; j < i; j++) sum -= MM(i, j) * x[j]; }
        else if (sum != 0.0) ii = i + 1;
        x[i] = sum;
    }
    for (int i = n - 1; i >= 0; i--)
    {
        double sum = x[i];
        for (int j = i + 1; j < n; j++) sum -= MM(i, j) * x[j];
        x[i] = sum / MM(i, i);
    }
#undef MM
    free(M); free(piv); free(vv);
}

/* GAMGSolverSolve.C:430-487 */
static void solve_coarsest(const orc_gamg* g, const orc_opts* o, double* corr, const double* src)
{
    const orc_sys* A = &g->lev[g->nLevels - 1].sys;
    if (o->directSolveCoarsest) { lu_direct_solve(A, corr, src); return; }
    for (int i = 0; i < A->nCellsTotal; i++) corr[i] = 0;
    orc_opts co;
    orc_default_opts(&co);
    co.tolerance = o->tolerance;
    co.relTol = o->relTol;
    int asym = 0;
    for (int d = 0; d < A->nDom; d++) if (A->dom[d].lower != A->dom[d].upper) asym = 1;
    if (asym) { co.solver = ORC_PBICG; co.precond = ORC_PRE_DILU; } /* BICCG */
    else      { co.solver = ORC_PCG;   co.precond = ORC_PRE_DIC;  } /* ICCG.C:46 */
    orc_solve(A, &co, corr, src, 0, 0);
}

/* GAMGSolverSolve.C:120-364 */
static void vcycle(const orc_gamg* g, const orc_opts* o, vcycle_ws* w, double* psi,
                   const double* source, double* Apsi, double* finestCorrection,
                   double* finestResidual)
{
    const int coarsestLevel = g->nLevels - 1;
    const int scaleCorrection = o->scaleCorrection < 0 ? g->sym : o->scaleCorrection;

    restrict_field(g, w->coarseSrc[0], finestResidual, 0);

    for (int leveli = 0; leveli < coarsestLevel; leveli++)
    {
        if (o->nPreSweeps)
        {
            const orc_sys* A = &g->lev[leveli].sys;
            const int n = A->nCellsTotal;
            for (int i = 0; i < n; i++) w->coarseCorr[leveli][i] = 0.0;
            orc_smooth(A, o->smoother, w->coarseCorr[leveli], w->coarseSrc[leveli],
                       imin(o->nPreSweeps + o->preSweepsLevelMultiplier * leveli, o->maxPreSweeps));
            double* ACf = Apsi; /* sub-field of Apsi */
            if (scaleCorrection && leveli < coarsestLevel - 1)
                gamg_scale(A, w->coarseCorr[leveli], ACf, w->coarseSrc[leveli]);
            orc_Amul(A, ACf, w->coarseCorr[leveli]);
            for (int i = 0; i < n; i++) w->coarseSrc[leveli][i] -= ACf[i];
        }
        restrict_field(g, w->coarseSrc[leveli + 1], w->coarseSrc[leveli], leveli + 1);
    }

    solve_coarsest(g, o, w->coarseCorr[coarsestLevel], w->coarseSrc[coarsestLevel]);

    for (int leveli = coarsestLevel - 1; leveli >= 0; leveli--)
    {
        const orc_sys* A = &g->lev[leveli].sys;
        const int n = A->nCellsTotal;
        double* preSmoothed = finestCorrection; /* sub-field of finestCorrection */
        if (o->nPreSweeps)
            for (int i = 0; i < n; i++) preSmoothed[i] = w->coarseCorr[leveli][i];

        prolong_field(g, w->coarseCorr[leveli], w->coarseCorr[leveli + 1], leveli + 1);

        double* ACf = Apsi;
        if (o->interpolateCorrection)
            gamg_interpolate(A, w->coarseCorr[leveli], ACf);
        if (scaleCorrection && leveli < coarsestLevel - 1)
            gamg_scale(A, w->coarseCorr[leveli], ACf, w->coarseSrc[leveli]);
        if (o->nPreSweeps)
            for (int i = 0; i < n; i++) w->coarseCorr[leveli][i] += preSmoothed[i];

        orc_smooth(A, o->smoother, w->coarseCorr[leveli], w->coarseSrc[leveli],
                   imin(o->nPostSweeps + o->postSweepsLevelMultiplier * leveli, o->maxPostSweeps));
    }

    prolong_field(g, finestCorrection, w->coarseCorr[0], 0);
    if (o->interpolateCorrection) gamg_interpolate(g->fine, finestCorrection, Apsi);
    if (scaleCorrection) gamg_scale(g->fine, finestCorrection, Apsi, finestResidual);
    for (int i = 0; i < g->fine->nCellsTotal; i++) psi[i] += finestCorrection[i];
    orc_smooth(g->fine, o->smoother, psi, source, o->nFinestSweeps);
}

static int check_convergence(orc_perf* p, double tol, double relTol)
{
    p->converged = (p->finalResidual < tol
        || (relTol > ORC_SMALL && p->finalResidual < relTol * p->initialResidual)) ? 1 : 0;
    return p->converged;
}

/* GAMGSolverSolve.C:34-117 */
orc_perf orc_gamg_solve(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                        const double* faceWeights, double* hist)
{
    orc_perf perf; memset(&perf, 0, sizeof(perf));
    orc_gamg* g = orc_gamg_build(s, o, faceWeights);
    if (!g || g->nLevels == 0)
    {
        /* GAMGSolver.C:108-126 FatalError "No coarse levels created" */
        fprintf(stderr, "orc_gamg_solve: no coarse levels created\n");
        perf.nIterations = -1;
        if (g) orc_gamg_free(g);
        return perf;
    }
    const int n = s->nCellsTotal;
    double* Apsi = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* finestCorrection = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    double* finestResidual = (double*)malloc(sizeof(double) * (size_t)(n + 1));

    orc_Amul(s, Apsi, psi);
    const double normFactor = orc_normFactor(s, psi, source, Apsi, finestCorrection);
    perf.normFactor = normFactor;
    for (int i = 0; i < n; i++) finestResidual[i] = source[i] - Apsi[i];
    perf.initialResidual = orc_gSumMag(s, finestResidual) / normFactor;
    perf.finalResidual = perf.initialResidual;
    if (hist) hist[perf.nHist] = perf.finalResidual;
    perf.nHist++;

    if (!check_convergence(&perf, o->tolerance, o->relTol))
    {
        vcycle_ws w = ws_new(g);
        do
        {
            vcycle(g, o, &w, psi, source, Apsi, finestCorrection, finestResidual);
            orc_Amul(s, Apsi, psi);
            for (int i = 0; i < n; i++) finestResidual[i] = source[i];
            for (int i = 0; i < n; i++) finestResidual[i] -= Apsi[i];
            perf.finalResidual = orc_gSumMag(s, finestResidual) / normFactor;
            if (hist) hist[perf.nHist] = perf.finalResidual;
            perf.nHist++;
        } while (++perf.nIterations < o->maxIter
                 && !check_convergence(&perf, o->tolerance, o->relTol));
        ws_free(g, &w);
    }
    free(Apsi); free(finestCorrection); free(finestResidual);
    orc_gamg_free(g);
    return perf;
}

/* ------------------------------------------------------------------
 * preconditioners/GAMGPreconditioner/GAMGPreconditioner.C:44-128 */
struct orc_gamg_pre {
    orc_gamg* g;
    orc_opts o;
    vcycle_ws w;
    double *AwA, *finestCorrection, *finestResidual;
};
typedef struct orc_gamg_pre orc_gamg_pre;

orc_gamg_pre* orc_gamg_pre_new(const orc_sys* s, const orc_opts* o, const double* faceWeights)
{
    orc_gamg_pre* p = (orc_gamg_pre*)calloc(1, sizeof(orc_gamg_pre));
    p->o = *o;
    p->g = orc_gamg_build(s, o, faceWeights);
    p->w = ws_new(p->g);
    const int n = s->nCellsTotal;
    p->AwA = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    p->finestCorrection = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    p->finestResidual = (double*)malloc(sizeof(double) * (size_t)(n + 1));
    return p;
}

void orc_gamg_pre_apply(orc_gamg_pre* p, double* wA, const double* rA)
{
    const orc_sys* s = p->g->fine;
    const int n = s->nCellsTotal;
    for (int i = 0; i < n; i++) wA[i] = 0.0;
    for (int i = 0; i < n; i++) p->finestResidual[i] = rA[i];
    for (int cycle = 0; cycle < p->o.nVcycles; cycle++)
    {
        vcycle(p->g, &p->o, &p->w, wA, rA, p->AwA, p->finestCorrection, p->finestResidual);
        if (cycle < p->o.nVcycles - 1)
        {
            orc_Amul(s, p->AwA, wA);
            for (int i = 0; i < n; i++) p->finestResidual[i] = rA[i];
            for (int i = 0; i < n; i++) p->finestResidual[i] -= p->AwA[i];
        }
    }
}

void orc_gamg_pre_free(orc_gamg_pre* p)
{
    if (!p) return;
    ws_free(p->g, &p->w);
    free(p->AwA); free(p->finestCorrection); free(p->finestResidual);
    orc_gamg_free(p->g);
    free(p);
}
