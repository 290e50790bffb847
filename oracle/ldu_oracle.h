/*
 * TEST INFRASTRUCTURE - CPU oracle for the lduMatrix solver hot path.
 *
 * A plain-C restatement of the reference's algorithms (OpenFOAM-2.2.x,
 * src/OpenFOAM/matrices/lduMatrix/...), every function citing the reference
 * file:line it follows (loop order preserved, compiled with -ffp-contract=off
 * so no FMA contraction, like the reference's x86-64 -O2/-O3 build).
 *
 * Pinned against the REAL reference (oracle/_ref/libOpenFOAM.so, built by
 * oracle/build_ref.sh) by tests/test_oracle_vs_ref.py and against the golden
 * vectors under tests/golden/ (generated from the real reference by
 * tests/golden/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (openfoam-2.2.x_amd/) never links or loads it.
 *
 * Multi-rank runs are emulated serially: a "system" is a list of sub-domains
 * (one per MPI rank of the reference) whose processor patches are paired
 * explicitly; halo values are snapshotted exactly where the reference posts
 * its sends (initMatrixInterfaces) and consumed where it waits
 * (updateMatrixInterfaces), reductions are summed in rank order.
 */
#ifndef LDU_ORACLE_H
#define LDU_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_patch {
    int n;                    /* faces on this coupled patch */
    const int* faceCells;     /* lduAddr().patchAddr(patchI) */
    const double* bouCoeffs;  /* interfaceBouCoeffs[patchI] */
    const double* intCoeffs;  /* interfaceIntCoeffs[patchI] */
    int nbrDom;               /* emulated neighbour rank */
    int nbrPatch;             /* patch index on the neighbour rank */
} orc_patch;

typedef struct orc_dom {
    int nCells, nFaces;
    const int* l;             /* lowerAddr (owner) */
    const int* u;             /* upperAddr (neighbour) */
    const double* diag;
    const double* upper;
    const double* lower;      /* == upper when symmetric (lduMatrix.C:198-215) */
    int nPatches;
    const orc_patch* patches;
    int cellOffset;           /* offset of this domain in concatenated vectors */
    /* derived addressing, built by orc_sys_finalize */
    int* losort;
    int* ownerStart;
    int* losortStart;
} orc_dom;

/* CPU baseline on all host cores (SURVEY.md 8d "CPU reference timing"): the sub-domains of a decomposed system are
 * the reference's MPI ranks; in the build with -fopenmp -DORC_OMP (libldu_oracle_omp.so) every loop over domains whose
 * iterations touch only their own domain's data runs one domain per thread.  Per-domain arithmetic and the rank-ordered
 * sums are untouched, so the results are bit-identical to the serial build (tests/test_multidomain_oracle.py). */
#ifdef ORC_OMP
#define ORC_PAR _Pragma("omp parallel for schedule(static, 1)")
#else
#define ORC_PAR
#endif
#define ORC_MAXDOM 256

typedef struct orc_sys {
    int nDom;
    orc_dom* dom;
    int nCellsTotal;
} orc_sys;

enum { ORC_PCG = 0, ORC_PBICG = 1, ORC_SMOOTH = 2, ORC_GAMG = 3, ORC_DIAGONAL = 4 };
enum { ORC_PRE_NONE = 0, ORC_PRE_DIAGONAL = 1, ORC_PRE_DIC = 2, ORC_PRE_FDIC = 3,
       ORC_PRE_DILU = 4, ORC_PRE_GAMG = 5 };
enum { ORC_SM_GS = 0, ORC_SM_SYMGS = 1, ORC_SM_DIC = 2, ORC_SM_DILU = 3,
       ORC_SM_DICGS = 4, ORC_SM_DILUGS = 5, ORC_SM_FDIC = 6, ORC_SM_NONBLOCKINGGS = 7 };
enum { ORC_AGG_FACEAREAPAIR = 0, ORC_AGG_ALGEBRAICPAIR = 1 };

typedef struct orc_opts {
    int solver, precond, smoother;
    double tolerance, relTol;     /* lduMatrixSolver.C:164-169 defaults 1e-6, 0 */
    int maxIter;                  /* default 1000 */
    int nSweeps;                  /* smoothSolver.C:73 default 1 */
    /* GAMGSolver.C:157-181 */
    int nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps;
    int nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps;
    int nFinestSweeps, interpolateCorrection, scaleCorrection /* -1 = matrix.symmetric() */;
    int nCellsInCoarsestLevel, mergeLevels, agglomerator;
    int nVcycles;                 /* GAMGPreconditioner.C:77 default 2 */
    int directSolveCoarsest;      /* GAMGSolver.C:76,180 default false: LU of the coarsest level (parallel: gathered over the ranks) */
} orc_opts;

typedef struct orc_perf {
    double initialResidual, finalResidual, normFactor;
    int nIterations, converged, singular;
    int nHist;                    /* residual history entries written */
} orc_perf;

void orc_default_opts(orc_opts* o);
void orc_sys_finalize(orc_sys* s);
void orc_sys_free_derived(orc_sys* s);

/* lduAddressing.C:31-169 */
void orc_calc_losort(int nCells, int nFaces, const int* u, int* losort);
void orc_calc_ownerStart(int nCells, int nFaces, const int* l, int* ownerStart);
void orc_calc_losortStart(int nCells, int nFaces, const int* u, const int* losort, int* losortStart);

/* lduMatrixATmul.C */
void orc_Amul(const orc_sys* s, double* Apsi, const double* psi);
void orc_Tmul(const orc_sys* s, double* Tpsi, const double* psi);
void orc_sumA(const orc_sys* s, double* sumA);
void orc_residual(const orc_sys* s, double* rA, const double* psi, const double* source);
/* lduMatrixTemplates.C:34-110, lduMatrixATmul.C:298-327 */
void orc_H(const orc_dom* d, double* H, const double* psi);
void orc_H1(const orc_dom* d, double* H1);
void orc_faceH(const orc_dom* d, double* faceH, const double* psi);

/* reductions: FieldM.H:385-394 serial left-to-right, then rank-order sum */
double orc_gSumProd(const orc_sys* s, const double* a, const double* b);
double orc_gSumMag(const orc_sys* s, const double* a);
double orc_normFactor(const orc_sys* s, const double* psi, const double* source,
                      const double* Apsi, double* tmp);

/* preconditioners (rank-local) */
void orc_DIC_calcReciprocalD(const orc_dom* d, double* rD);
void orc_DILU_calcReciprocalD(const orc_dom* d, double* rD);
void orc_DIC_precondition(const orc_dom* d, const double* rD, double* wA, const double* rA);
void orc_FDIC_precondition(const orc_dom* d, const double* rD, const double* rDuUpper,
                           const double* rDlUpper, double* wA, const double* rA);
void orc_DILU_precondition(const orc_dom* d, const double* rD, double* wA, const double* rA);
void orc_DILU_preconditionT(const orc_dom* d, const double* rD, double* wT, const double* rT);

/* smoothers (system level: halo snapshot per sweep) */
void orc_smooth(const orc_sys* s, int smoother, double* psi, const double* source, int nSweeps);

/* solvers; resHist must hold maxIter+2 doubles (may be NULL) */
orc_perf orc_solve(const orc_sys* s, const orc_opts* o, double* psi, const double* source,
                   const double* faceWeights, double* resHist);

/* GAMG agglomeration (single domain): returns an opaque hierarchy */
typedef struct orc_gamg orc_gamg;
orc_gamg* orc_gamg_build(const orc_sys* s, const orc_opts* o, const double* faceWeights);
void orc_gamg_free(orc_gamg* g);
int orc_gamg_nLevels(const orc_gamg* g);
int orc_gamg_level_nCells(const orc_gamg* g, int lev);
int orc_gamg_level_nCells_dom(const orc_gamg* g, int lev, int d);
int orc_gamg_level_nFaces_dom(const orc_gamg* g, int lev, int d);
int orc_gamg_level_nFaces(const orc_gamg* g, int lev);
const int* orc_gamg_restrict(const orc_gamg* g, int lev);
const int* orc_gamg_faceRestrict(const orc_gamg* g, int lev);
const int* orc_gamg_level_lower(const orc_gamg* g, int lev);
const int* orc_gamg_level_upper(const orc_gamg* g, int lev);
const double* orc_gamg_level_diag(const orc_gamg* g, int lev);
const double* orc_gamg_level_upperCoeffs(const orc_gamg* g, int lev);
const double* orc_gamg_level_lowerCoeffs(const orc_gamg* g, int lev);

#ifdef __cplusplus
}
#endif
#endif
