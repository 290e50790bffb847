/*
 * ldugpu.h - C ABI of the MI355X-native lduMatrix solver hot path.
 *
 * Drop-in boundary = OpenFOAM-2.2.x's lduMatrix::solver / preconditioner /
 * smoother interface (src/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrix.H:91-506).
 * The OpenFOAM-side shim (openfoam-2.2.x_amd/plugin/hipLduSolvers.C) and the
 * Python host mirror (openfoam-2.2.x_amd/capi.py, tests and bench) are the only callers; both hand over
 * exactly what the reference hands to its solvers (SURVEY.md 8b "data handed over"):
 * raw contiguous f64 coefficient arrays, int32 addressing, psi and source.
 *
 * Plain pointers and sizes only.  Pointers may be host or device pointers
 * (detected with hipPointerGetAttributes); device pointers avoid the PCIe copy.
 * All functions return 0 on success or a negative error code; ldu_last_error()
 * returns the message (the OpenFOAM shim turns it into FatalErrorIn(...)).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef LDUGPU_H
#define LDUGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ldu_ctx ldu_ctx;       /* device context: stream, scratch, communicator   */
typedef struct ldu_addr ldu_addr;     /* device image of one lduAddressing (+ coupled patches) */
typedef struct ldu_matrix ldu_matrix; /* device image of one lduMatrix (coefficients)    */

/* ---- enums (names = the reference's run-time-selection keys) ------------------- */
enum { LDU_SOLVER_PCG = 0, LDU_SOLVER_PBICG = 1, LDU_SOLVER_SMOOTH = 2, LDU_SOLVER_GAMG = 3,
       LDU_SOLVER_DIAGONAL = 4 };
enum { LDU_PRE_NONE = 0, LDU_PRE_DIAGONAL = 1, LDU_PRE_DIC = 2, LDU_PRE_FDIC = 3, LDU_PRE_DILU = 4,
       LDU_PRE_GAMG = 5 };
enum { LDU_SM_GAUSSSEIDEL = 0, LDU_SM_SYMGAUSSSEIDEL = 1, LDU_SM_DIC = 2, LDU_SM_DILU = 3,
       LDU_SM_DICGAUSSSEIDEL = 4, LDU_SM_DILUGAUSSSEIDEL = 5, LDU_SM_FDIC = 6,
       LDU_SM_NONBLOCKINGGAUSSSEIDEL = 7 /* == GaussSeidel bit-for-bit when no coupled patch is present */ };
enum { LDU_AGG_FACEAREAPAIR = 0, LDU_AGG_ALGEBRAICPAIR = 1 };

/* Solver controls = the keys the reference reads from the fvSolution sub-dictionary:
 * lduMatrixSolver.C:164-169 (maxIter 1000, tolerance 1e-6, relTol 0), smoothSolver.C:73
 * (nSweeps), GAMGSolver.C:157-181, GAMGAgglomeration.C:79 / pairGAMGAgglomeration.C:45,
 * GAMGPreconditioner.C:77 (nVcycles). */
typedef struct ldu_controls {
    int32_t solver, preconditioner, smoother;
    double tolerance, relTol;
    int32_t maxIter;
    int32_t nSweeps;
    int32_t cacheAgglomeration;
    int32_t nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps;
    int32_t nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps;
    int32_t nFinestSweeps;
    int32_t interpolateCorrection;
    int32_t scaleCorrection;      /* -1 = matrix.symmetric() (GAMGSolver.C:74) */
    int32_t directSolveCoarsest;  /* GAMGSolver.C:76,95-106: LU of the coarsest level (<= 64 cells; with coupled patches / several ranks the gathered
                                     matrix of LUscalarMatrix.C:52-107, <= 128 cells over all ranks; else refused) */
    int32_t nCellsInCoarsestLevel, mergeLevels, agglomerator;
    int32_t nVcycles;
    int32_t historyCapacity;      /* residual history entries the caller's buffer holds */
} ldu_controls;

/* SolverPerformance<scalar> (src/OpenFOAM/matrices/LduMatrix/LduMatrix/SolverPerformance.H) */
typedef struct ldu_perf {
    double initialResidual, finalResidual, normFactor;
    int32_t nIterations, converged, singular;
    int32_t nHistory;             /* residual-history entries written */
    double solveSeconds;          /* wall time inside the solve (device-synchronised) */
    double setupSeconds;          /* coefficient upload / preconditioner / level setup */
} ldu_perf;

const char* ldu_last_error(void);
void ldu_default_controls(ldu_controls* c);

/* ---- context ------------------------------------------------------------------- */
/* One context per rank/GPU (reference: one MPI rank per sub-domain, UPstream). */
int ldu_ctx_create(ldu_ctx** ctx, int device);
int ldu_ctx_destroy(ldu_ctx* ctx);
int ldu_ctx_sync(ldu_ctx* ctx);
/* The sequential sweeps of the reference (DICPreconditioner.C:87-123, GaussSeidelSmoother.C:66-187) always
 * complete.  Here the point-to-point / cluster sweep engines bound every dependency wait (`polls` granule
 * polls, default 2^22, env LDU_SPIN_LIMIT); an operation in which a wait expired is re-run from its inputs on
 * the level-kernel engine (bit-identical results) instead of failing.  ldu_ctx_fallback_count = how many
 * operations of this context took that path.  polls = 0 restores the default. */
int ldu_ctx_set_spin_limit(ldu_ctx* ctx, uint32_t polls);
/* Time bound of the dependency waits of the sweep engines (besides the poll bound of ldu_ctx_set_spin_limit): a wave
 * that has waited longer than budgetMs (wall clock; default 200, 0 = no time bound) gives up, the sweep drains and the
 * operation is re-run on the level-kernel engine (ldu_ctx_fallback_count counts) - a launch that crawls cannot stall a
 * solve.  debugStallMs > 0 (tests only) makes the wave that runs the first task of every sweep launch sit still for
 * that long.  The value is a device variable per kernel file: it applies to every context ON THE DEVICE OF `ctx` (a
 * process that drives several devices sets it once per device).  A context whose fast engines gave up three operations
 * in a row (a GPU shared with other processes, a profiler or a debugger can push healthy launches past the budget) keeps
 * the level-kernel engine from then on instead of paying the failed attempt and the re-run on every operation.
 * Multi-rank contract: ldu_smooth, ldu_precondition, ldu_solve and the scalar read-backs max-reduce the abort flag over
 * the ranks before reading it, i.e. they are COLLECTIVE - every rank of the communicator must make the same sequence of
 * these calls (as every rank of an OpenFOAM run does), a rank that skips one stalls the others. */
int ldu_ctx_set_watchdog(ldu_ctx* ctx, double budgetMs, double debugStallMs);
/* communication counters of a context since its creation: out[0] halo exchanges with other ranks (initMatrixInterfaces),
 * [1] scalar all-reduces (reduce(..., sumOp)), [2] device-to-host read-backs of solver scalars (one per convergence
 * check), [3] halo exchanges that ran on the communication stream overlapped with the interior rows */
int ldu_ctx_comm_counters(const ldu_ctx* ctx, int64_t out[4]);
int64_t ldu_ctx_fallback_count(const ldu_ctx* ctx);
/* RCCL halo exchanges that ran on the communication stream, overlapped with the interior rows of the operator
 * (between initMatrixInterfaces and updateMatrixInterfaces, lduMatrixUpdateMatrixInterfaces.C:30-93, 127-160);
 * LDU_HALO_OVERLAP=0 puts them back on the compute stream. */
int64_t ldu_ctx_overlapped_halo_count(const ldu_ctx* ctx);
/* Multi-GPU: RCCL communicator from a 128-byte unique id shared out-of-band
 * (replaces UPstream::init / MPI_COMM_WORLD, src/Pstream/mpi/UPstream.C). */
int ldu_comm_unique_id(uint8_t id[128]);
int ldu_ctx_comm_init(ldu_ctx* ctx, int rank, int nRanks, const uint8_t id[128]);
/* Test facility for 1-GPU boxes: nRanks contexts of ONE process (one host thread each) form a
 * local group that exchanges halos / reductions through device copies instead of RCCL. */
int ldu_ctx_comm_init_local(ldu_ctx* ctx, int rank, int nRanks, int groupId);
/* Peer-store backend (intra-node, xGMI): every rank owns a window of fine-grained device memory that all other ranks
 * map (hipIpc across processes); processor-patch values (lduMatrixUpdateMatrixInterfaces.C:30-160) and the partial sums
 * of reduce(scalar, sumOp) (FieldFunctions.C:514-533) are STORED into the neighbour's window by the producing kernel
 * and polled by the consuming one - no collective library, no extra launches in the steady state, sums formed in rank
 * order on every rank.  The few set-up-time messages (window handles, receive offsets, restrict maps of
 * processorGAMGInterface.C:137-154, the and-reduce of GAMGAgglomeration.C:53-62) go through `oob`, a pairwise exchange
 * the HOST application provides on its own transport (OpenFOAM: Pstream; Python tests: torch.distributed / gloo):
 *   oob(user, nPeers, peers[], sendBufs[], sendBytes[], recvBufs[], recvBytes[]) sends sendBytes[i] bytes to rank
 *   peers[i] and receives recvBytes[i] bytes from it, for all i, and returns 0; every rank of a pair makes the
 *   matching call (sizes agree by construction).  nRanks = 1 needs no callback.
 * With an RCCL communicator on the same context (ldu_ctx_comm_init, either order) RCCL stays the default carrier and
 * LDU_HALO=p2p / LDU_REDUCE=p2p or ldu_ctx_comm_select move the halo exchanges / the global sums to peer stores;
 * without one everything travels by peer stores.  LDU_PEER_WINDOW_MB (256), LDU_PEER_TIMEOUT_S (20). */
typedef int (*ldu_oob_exchange_fn)(void* user, int32_t nPeers, const int32_t* peers, const void* const* sendBufs,
                                   const int64_t* sendBytes, void* const* recvBufs, const int64_t* recvBytes);
int ldu_ctx_comm_init_peer(ldu_ctx* ctx, int rank, int nRanks, ldu_oob_exchange_fn oob, void* user);
int ldu_ctx_comm_select(ldu_ctx* ctx, int peerHalo, int peerReduce);
/* out[0] ranks of the RCCL communicator (ncclCommCount; 0 = none), [1] ranks whose windows are mapped (0 = no peer
 * backend), [2] / [3] = 1 when the halo exchanges / the global sums travel by peer stores */
int ldu_ctx_comm_info(const ldu_ctx* ctx, int32_t out[4]);

/* ---- addressing (lduAddressing.H:111-199, lduPrimitiveMesh.H:83-99) ------------ */
/* lower/upper = lowerAddr()/upperAddr(), upper-triangular order (sorted by owner).  */
int ldu_addr_create(ldu_ctx* ctx, ldu_addr** a, int32_t nCells, int32_t nFaces,
                    const int32_t* lowerAddr, const int32_t* upperAddr);
/* Coupled (processor) patch: lduAddr().patchAddr(patchI) = faceCells; neighbour rank
 * (processorLduInterface::neighbProcNo).  Patches must be added in patch order. */
int ldu_addr_add_patch(ldu_addr* a, int32_t nPatchFaces, const int32_t* faceCells, int32_t nbrRank);
/* cyclic coupled patch (cyclicLduInterface, src/OpenFOAM/matrices/lduMatrix/lduAddressing/lduInterface/
 * cyclicLduInterface.H): the neighbour is patch `nbrPatch` (index in order of addition, may be added
 * later) of the SAME addressing; face i pairs with face i of the neighbour patch.  Untransformed
 * coupling only (scalars / translational cyclics: cyclicFvPatchField::updateInterfaceMatrix with
 * doTransform() false). */
int ldu_addr_add_cyclic_patch(ldu_addr* a, int32_t nFaces, const int32_t* faceCells, int32_t nbrPatch);
/* Call after the last add_patch (builds the device-side schedule). */
int ldu_addr_finalize(ldu_addr* a);
int ldu_addr_destroy(ldu_addr* a);
/* Diagnostics: dependency levels of the triangular sweeps etc. */
int ldu_addr_info(const ldu_addr* a, int32_t* nLevels, int32_t* nSlices, int64_t* nEntriesPadded);
/* which sweep engine serves kind (0 triangular DIC/DILU sweeps, 1 one GaussSeidel sweep, 2 two pipelined
 * GaussSeidel sweeps) on this addressing: 0 chip-wide point-to-point, 1 XCD slabs, 2 clusters, 3 single
 * wavefront, 4 level kernels.  Measurement / test introspection. */
int ldu_addr_sweep_engine(ldu_addr* a, int32_t kind);
/* Face weights for the geometric agglomerator (faceAreaPairGAMGAgglomeration.C:48-73
 * computes them from Sf; the shim passes mag(cmptMultiply(Sf/sqrt(magSf),(1,1.01,1.02)))). */
int ldu_addr_set_face_weights(ldu_addr* a, const double* faceWeights);
/* Sub-domain mode: K ranks of the reference inside ONE addressing.  The caller hands over the matrices of the K ranks
 * concatenated (cells of rank 0, rank 1, ...; processor patch (a -> b) as a cyclic patch paired with (b -> a),
 * ldu_addr_add_cyclic_patch) and says here which cell belongs to which rank.  Every operator then IS the K-rank operator
 * (GaussSeidelSmoother.C:98-145: neighbour ranks' values of the previous sweep; lduMatrixUpdateMatrixInterfaces.C:30-160),
 * computed in one set of launches, and GAMG coarsens rank by rank as the K-rank run does: pairs never cross an interface,
 * and coarsening stops when ANY sub-domain would fall below nCellsInCoarsestLevel (GAMGAgglomeration.C:53-62, the
 * and-reduce of continueAgglomerating).  Call before the first solve; nSub = 0 switches it off. */
int ldu_addr_set_subdomains(ldu_addr* a, int32_t nSub, const int32_t* cellSub);
/* The same from the face area vectors themselves, as faceAreaPairGAMGAgglomeration.C:48-73 does with
 * fvMesh::Sf() / magSf() (= mag(Sf) + VSMALL, fvMeshGeometry.C:101-114): Sf = nFaces*3 doubles (internal
 * faces = the first nInternalFaces entries of primitiveMesh::faceAreas()), host or device.  The weights are
 * computed on the device; ldu_addr_get_face_weights returns them (host or device destination). */
int ldu_addr_set_face_areas(ldu_addr* a, const double* Sf);
int ldu_addr_get_face_weights(const ldu_addr* a, double* faceWeights);
/* number of HIP devices visible to this process (rank -> GPU binding of the shim) */
int ldu_device_count(void);

/* ---- matrix (lduMatrix.H:82-85; symmetric <=> lower == NULL, lduMatrix.C:198-215) -- */
int ldu_matrix_create(ldu_addr* a, ldu_matrix** m);
int ldu_matrix_destroy(ldu_matrix* m);
/* Re-read every solve (fvScalarMatrix.C:152-174 modifies diag around the call). */
int ldu_matrix_set_coeffs(ldu_matrix* m, const double* diag, const double* upper,
                          const double* lower /* NULL = symmetric */);
/* interfaceBouCoeffs[patchI] / interfaceIntCoeffs[patchI] of coupled patch patchI */
int ldu_matrix_set_patch_coeffs(ldu_matrix* m, int32_t patchI, const double* bouCoeffs,
                                const double* intCoeffs);

/* ---- matrix operations on caller vectors (original cell order) -------------------- */
/* lduMatrix::Amul / Tmul / sumA / residual (lduMatrixATmul.C:34-280) */
int ldu_amul(ldu_matrix* m, double* Apsi, const double* psi);
int ldu_tmul(ldu_matrix* m, double* Tpsi, const double* psi);
int ldu_sumA(ldu_matrix* m, double* sumA);
int ldu_residual(ldu_matrix* m, double* rA, const double* psi, const double* source);
/* lduMatrix::H / H1 / faceH (lduMatrixTemplates.C:34-110, lduMatrixATmul.C:298-327) */
int ldu_H(ldu_matrix* m, double* H, const double* psi);
int ldu_H1(ldu_matrix* m, double* H1);
int ldu_faceH(ldu_matrix* m, double* faceH, const double* psi);
/* gSumProd / gSumMag (FieldFunctions.C:477-503): rank-local device reduction + RCCL all-reduce */
int ldu_gSumProd(ldu_matrix* m, const double* a, const double* b, double* result);
int ldu_gSumMag(ldu_matrix* m, const double* a, double* result);

/* lduMatrix::preconditioner::precondition / preconditionT (lduMatrix.H:482-505) */
int ldu_precondition(ldu_matrix* m, int32_t preconditioner, double* wA, const double* rA,
                     int32_t transpose);
/* lduMatrix::smoother::smooth (lduMatrix.H:391-397) */
int ldu_smooth(ldu_matrix* m, int32_t smoother, double* psi, const double* source, int32_t nSweeps);

/* ---- whole solvers: lduMatrix::solver::solve (lduMatrix.H:242-247) ------------------ */
/* psi in/out, source in; resHistory (may be NULL) receives the residual after every
 * checkConvergence call like the reference's debug>=2 print (SolverPerformance.C:65-71). */
int ldu_solve(ldu_matrix* m, const ldu_controls* controls, double* psi, const double* source,
              ldu_perf* perf, double* resHistory);

/* GAMG hierarchy introspection (tests): level sizes and restrict maps */
int ldu_gamg_levels(ldu_matrix* m, const ldu_controls* controls, int32_t* nLevels,
                    int32_t* nCellsPerLevel /* [50] */, int32_t* nFacesPerLevel /* [50] */);
/* measurement introspection of a built hierarchy: info = {nCells, nFaces, dependency levels, widest row,
 * engine of the triangular sweeps, of one GaussSeidel sweep, of pipelined GaussSeidel sweeps (codes of
 * ldu_addr_sweep_engine), slices} of coarse level `level` (0 = first coarse level) */
int ldu_gamg_level_info(ldu_matrix* m, int32_t level, int32_t info[8]);
int ldu_gamg_level_data(ldu_matrix* m, int32_t level, int32_t* restrictAddr /* fine nCells */,
                        double* diag, double* upper, double* lower /* may be NULL */);

/* ---- measurement: HIP-event timing of kernel classes on the library's compute stream -----
 * (the stream is private to the library, so an outside torch.cuda.Event cannot see it).
 * Only operations on the addressing of `m` (the finest level) are recorded. */
enum { LDU_PROF_AMUL = 0,      /* Amul/Tmul row kernel: one launch each                      */
       LDU_PROF_GS_SWEEP = 1,  /* one GaussSeidel sweep = one graph launch of level kernels   */
       LDU_PROF_TRI_SWEEP = 2, /* one DIC/DILU forward or backward sweep (graph launch)       */
       LDU_PROF_RESIDUAL = 3,  /* residual row kernel                                         */
       LDU_PROF_NCATS = 8 };
int ldu_profile_begin(ldu_matrix* m);
/* ms[c] = total milliseconds, counts[c] = recorded launches of class c since begin */
int ldu_profile_end(ldu_matrix* m, double ms[8], int64_t counts[8]);

/* Debug: per-slice trace of the point-to-point sweep engine (8 x u64 per slice of the addressing of
 * m: tTicket, tWaitStart, tReady, tDone [shader clocks], polls, XCC id, workgroup, ticket).
 * buf = device memory of nSlices*64 bytes, or NULL to switch tracing off. */
int ldu_debug_p2p_trace(ldu_matrix* m, void* buf);
/* Debug: the first dependency wait that expired in an aborted sweep (row, tag, columns, seen tags). */
int ldu_debug_p2p_stuck(ldu_matrix* m, int32_t out[16]);
/* Debug: per-task timeline of the pipelined GaussSeidel sweeps of the cluster engine.  buf = device memory of
 * nSweeps * nClusters * 64 bytes (8 x u64 per (sweep, cluster): start, upper values there, lower values there,
 * steps done, stores acknowledged [100 MHz wall clock], polls, XCC id, workgroup), or NULL to switch tracing off.
 * ldu_debug_cluster_levels: out[0] = clusters, out[1] = cluster levels, out[2..] = first cluster of every level. */
int ldu_debug_cluster_trace(ldu_matrix* m, void* buf);
/* The same for the pipelined sweeps of the level engines: 8 x u64 per (sweep, slice) - start, upper values there,
 * lower values there, stored [100 MHz wall clock], XCC id, workgroup; buf = nSweeps * nSlices * 64 bytes of device
 * memory or NULL.  ldu_debug_slice_levels: out[0] = dependency levels, out[1..] = first slice of every level. */
int ldu_debug_gs_multi_trace(ldu_matrix* m, void* buf);
int ldu_debug_slice_levels(ldu_matrix* m, int32_t* out, int32_t cap);
/* The block engine (ldu_blocks.hip): per (sweep, group) task 8 x u64 - start, loads issued, dependencies seen, stored
 * [100 MHz wall clock], wavefront, sweep, block, lanes per row; buf = (tasks of k sweeps) * 64 bytes or NULL.
 * ldu_debug_blocks_info: out[8]; out[0..5] = blocks, compute wavefronts per block, LDS bytes per block, ghosts, groupings,
 * tasks of k sweeps, out[6..7] reserved (written as 0) (all 0: the addressing is not on the block engine). */
/* The sweep plans of large GAMG levels (block-engine layouts, task orders of pipelined sweeps) are built on host threads behind
 * the first solves - the level engines sweep meanwhile, with bit-identical results; this waits until all of them are there
 * (benchmarks: before a timed region).  The engine queries (ldu_addr_sweep_engine, ldu_gamg_level_info) wait as well. */
int ldu_matrix_wait_plans(ldu_matrix* m);
int ldu_debug_blocks_trace(ldu_matrix* m, void* buf);
int ldu_debug_blocks_info(ldu_matrix* m, int32_t k, int64_t* out);
/* Per-sweep layouts of the chip-wide pipelined GaussSeidel sweeps (ldu_gslayouts.cpp): out[0] = sweeps with a layout of their own
 * built so far (0: none / not on this addressing), out[1 + j] = slices of sweep j's layout (j = 1 .. 3), out[5] = slices of the
 * level layout. */
int ldu_debug_gs_layouts(ldu_matrix* m, int64_t* out);
int ldu_debug_cluster_levels(ldu_matrix* m, int32_t* out, int32_t cap);
/* ldu_debug_slices: per slice of the level-ordered layout {first row, rows, entries per row, most lower, most upper
 * neighbours of a row}; out holds 5 * cap values, cap >= slices */
int ldu_debug_slices(ldu_matrix* m, int32_t* out, int32_t cap);
/* Debug: the GaussSeidel rows end in `curPsi /= diagPtr[cellI]` (GaussSeidelSmoother.C:154); the sweep kernels do the
 * denominator's half of that IEEE division ahead of the dependency wait.  This runs n operand pairs (random bit
 * patterns, exponents at the edges of the fast range, zeros, denormals, huge values) through that path and through
 * the compiler's division and returns the number of quotients that differ in any bit (must be 0). */
int ldu_debug_div_check(ldu_ctx* ctx, uint64_t seed, int64_t n, uint64_t* mismatches);
/* Achievable HBM bandwidth of this device with the library's own stream kernels (McCalpin STREAM, f64): mode 0 copy
 * a = b (16 B per element), mode 1 triad a = b + s*c (24 B per element); n doubles per array; *seconds = average
 * launch time over `reps` launches (HIP events on the library's stream).  bench.py: roofline.peak_measured. */
int ldu_debug_stream(ldu_ctx* ctx, int32_t mode, int64_t n, int32_t reps, double* seconds);

/* ---- finite-volume stencils feeding the matrix ------------- */
typedef struct ldu_mesh_geom {
    /* internal faces */
    const double* Sf;        /* [nFaces*3] face area vectors (fvMesh::Sf) */
    const double* magSf;     /* [nFaces]   */
    const double* weights;   /* [nFaces] linear interpolation weights (surfaceInterpolation.C:175-185) */
    const double* deltaCoeffs; /* [nFaces] (surfaceInterpolation.C:239-242) */
    const double* V;         /* [nCells] cell volumes */
} ldu_mesh_geom;
/* surfaceInterpolationScheme::interpolate (surfaceInterpolationScheme.C:293-296):
 * sf[f] = lambda[f]*(vf[own]-vf[nei]) + vf[nei]; nComp = 1 (scalar) or 3 (vector) */
int ldu_fv_interpolate(ldu_addr* a, int32_t nComp, const double* lambdas, const double* vf, double* sf);
/* fvc::surfaceIntegrate internal-face part (fvcSurfaceIntegrate.C:43-76), divided by V */
int ldu_fvc_surfaceIntegrate(ldu_addr* a, int32_t nComp, const double* ssf, const double* V,
                             double* ivf);
/* fv::gaussGrad::gradf internal part (gaussGrad.C:41-110): grad = sum_f Sf*ssf / V */
int ldu_fvc_gaussGrad(ldu_addr* a, const double* Sf, const double* ssf, const double* V, double* grad);
/* snGradScheme::snGrad (snGradScheme.C:139-143): ssf[f] = delta[f]*(vf[nei]-vf[own]) */
int ldu_fvc_snGrad(ldu_addr* a, const double* deltaCoeffs, const double* vf, double* ssf);
/* gaussLaplacianScheme::fvmLaplacianUncorrected (gaussLaplacianScheme.C:46-88):
 * upper = deltaCoeffs*gammaMagSf; diag = negSumDiag */
int ldu_fvm_laplacian(ldu_addr* a, const double* deltaCoeffs, const double* gammaMagSf,
                      double* diag, double* upper);
/* gaussConvectionScheme::fvmDiv (gaussConvectionScheme.C:68-107): lower = -w*phi;
 * upper = lower + phi; negSumDiag */
int ldu_fvm_div(ldu_addr* a, const double* weights, const double* faceFlux, double* diag,
                double* upper, double* lower);

/* ---- fvMatrix glue executed around every solve, scalar matrices ------------
 * Replaces the O(nCells + nBoundaryFaces) host loops of
 *   src/finiteVolume/fvMatrices/fvMatrix/fvMatrix.C: addBoundaryDiag :116-131, addBoundarySource
 *   :150-178, setReference :509-521, relax :525-655, A :722-746, flux :865-943; H = the scalar
 *   specialisation fvScalarMatrix.C:209-237
 * so that diag/source/psi stay in HBM between assembly and ldu_solve.  Cells and faces in the caller's
 * (original) numbering; pointers host or device.
 * A boundary = the patches of the fvMesh in order: patchSizes[nPatches], faceCells of all patches
 * concatenated (lduAddr().patchAddr(patchI)), coupled[patchI] = fvPatchField::coupled().  Per-patch
 * coefficient arrays (internalCoeffs_, boundaryCoeffs_, patchNeighbourField) are passed concatenated in
 * the same order. */
typedef struct ldu_fv_boundary ldu_fv_boundary;
int ldu_fv_boundary_create(ldu_addr* a, int32_t nPatches, const int32_t* patchSizes, const int32_t* faceCells,
                           const int32_t* coupled /* may be NULL: none */, ldu_fv_boundary** out);
int ldu_fv_boundary_destroy(ldu_fv_boundary* b);
/* diag[faceCells] += internalCoeffs  (fvMatrix.C:116-131) */
int ldu_fvm_addBoundaryDiag(ldu_fv_boundary* b, const double* internalCoeffs, double* diag);
/* source[faceCells] += boundaryCoeffs (non-coupled) | boundaryCoeffs*patchNeighbourField (coupled, if couples) */
int ldu_fvm_addBoundarySource(ldu_fv_boundary* b, const double* boundaryCoeffs, const double* patchNeighbourField,
                              int32_t couples, double* source);
/* fvMatrix<scalar>::relax(alpha): diag and source updated in place; lower NULL = symmetric */
int ldu_fvm_relax(ldu_fv_boundary* b, double alpha, const double* internalCoeffs, const double* boundaryCoeffs,
                  const double* upper, const double* lower, const double* psi, double* diag, double* source);
/* source[celli] += diag[celli]*value; diag[celli] += diag[celli]   (celli < 0: no-op) */
int ldu_fvm_setReference(ldu_addr* a, int32_t celli, double value, double* diag, double* source);
/* A = (diag + boundary diag)/V */
int ldu_fvm_A(ldu_fv_boundary* b, const double* internalCoeffs, const double* diag, const double* V, double* A);
/* H = (lduMatrix::H(psi) + source + boundary source)/V */
int ldu_fvm_H(ldu_fv_boundary* b, const double* internalCoeffs, const double* boundaryCoeffs,
              const double* patchNeighbourField, const double* upper, const double* lower, const double* psi,
              const double* source, const double* V, double* H);
/* flux: internal faces = lduMatrix::faceH(psi); boundary faces (concatenated) =
 * internalCoeffs*psi[faceCells] - boundaryCoeffs[*patchNeighbourField on coupled patches] */
int ldu_fvm_flux(ldu_fv_boundary* b, const double* internalCoeffs, const double* boundaryCoeffs,
                 const double* patchNeighbourField, const double* upper, const double* lower, const double* psi,
                 double* fluxInternal, double* fluxBoundary);

/* vector (3-component) matrices, fvMatrix<vector>: scalar diag/upper/lower, vector psi / source ([nCells][3])
 * and per-patch vector coefficients ([nPatchFaces][3], concatenated as above).  Same reference loops with
 * Type = vector: component(.,cmpt), cmptMultiply, cmptAv, cmptMax(cmptMag(.)), cmptMin. */
int ldu_fvm_addBoundaryDiagCmpt(ldu_fv_boundary* b, const double* internalCoeffs3, int32_t cmpt, double* diag);
int ldu_fvm_addBoundarySourceV(ldu_fv_boundary* b, const double* boundaryCoeffs3, const double* patchNeighbourField3,
                               int32_t couples, double* source3);
int ldu_fvm_relaxV(ldu_fv_boundary* b, double alpha, const double* internalCoeffs3, const double* boundaryCoeffs3,
                   const double* upper, const double* lower, const double* psi3, double* diag, double* source3);
int ldu_fvm_AV(ldu_fv_boundary* b, const double* internalCoeffs3, const double* diag, const double* V, double* A);
/* the generic fvMatrix<Type>::H (fvMatrix.C:751-813), not the scalar specialisation */
int ldu_fvm_HV(ldu_fv_boundary* b, const double* internalCoeffs3, const double* boundaryCoeffs3,
               const double* patchNeighbourField3, const double* upper, const double* lower, const double* psi3,
               const double* source3, const double* V, double* H3);

/* ---- higher-order schemes of the motorBike set-up, internal field ----------
 * linearUpwind<scalar>::correction (linearUpwind.C:87-91): corr[f] = (Cf[f] - C[c]) & gradVf[c],
 * c = faceFlux[f] > 0 ? owner : neighbour.  C3 [nCells][3], Cf3 [nFaces][3], gradVf3 [nCells][3]. */
int ldu_fv_linearUpwindCorrection(ldu_addr* a, const double* faceFlux, const double* C3, const double* Cf3,
                                  const double* gradVf3, double* corr);
/* cellLimitedGrad<scalar>::calcGrad (cellLimitedGrads.C:46-196): limits grad3 (in/out, the basic scheme's
 * gradient) so that the face extrapolates stay within the min/max of the neighbouring values; k as in
 * `cellLimited Gauss linear k`.  boundaryValues / boundaryCf3: per patch face (concatenated, order of
 * ldu_fv_boundary_create) the patch value - patchNeighbourField on coupled patches - and the face centre;
 * b may be NULL (no boundary faces). */
int ldu_fvc_cellLimitedGrad(ldu_addr* a, ldu_fv_boundary* b, double k, const double* vsf, const double* boundaryValues,
                            const double* C3, const double* Cf3, const double* boundaryCf3, double* grad3);
/* vector forms (what motorBike's fvSchemes selects for U: `div(phi,U) bounded Gauss linearUpwindV grad(U)`,
 * `grad(U) cellLimited Gauss linear 1`): linearUpwindV<vector>::correction (linearUpwindV.C:87-140, internal
 * faces; weights = linear weights, gradVf9 = [nCells][9] tensors xx xy xz yx ...) and
 * cellLimitedGrad<vector>::calcGrad (cellLimitedGrads.C:200-360; grad9 in/out) */
/* fv::gaussGrad<Type>::gradf with the patch faces (gaussGrad.C:41-110): nComp = 1 -> grad[nCells][3],
 * nComp = 3 (vector field) -> grad[nCells][9] (xx xy xz yx ..., component 3i+j = d_i U_j); ssf = the face values
 * (e.g. from ldu_fv_interpolate), boundarySf3 / boundarySsf concatenated in patch order */
int ldu_fvc_gaussGradFull(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* Sf3, const double* ssf,
                          const double* boundarySf3, const double* boundarySsf, const double* V, double* grad);
/* the `bounded` wrapper of those div schemes (boundedConvectionScheme.C:60-77): fvmDiv - fvm::Sp(fvc::
 * surfaceIntegrate(phi), vf), i.e. diag -= V*surfaceIntegrate(phi) (fvcSurfaceIntegrate.C:43-76; boundaryFlux:
 * the patch values of phi concatenated in patch order, b may be NULL for a mesh without patches) */
int ldu_fvm_boundedSp(ldu_addr* a, ldu_fv_boundary* b, const double* faceFlux, const double* boundaryFlux,
                      const double* V, double* diag);
int ldu_fv_linearUpwindVCorrection(ldu_addr* a, const double* faceFlux, const double* weights, const double* vf3,
                                   const double* C3, const double* Cf3, const double* gradVf9, double* corr3);
int ldu_fvc_cellLimitedGradV(ldu_addr* a, ldu_fv_boundary* b, double k, const double* vsf3, const double* boundaryValues3,
                             const double* C3, const double* Cf3, const double* boundaryCf3, double* grad9);

/* ---- non-orthogonal correction, gaussDiv, patch halves ---------------
 * What motorBike's fvSchemes adds to the uncorrected stencils above (`laplacianSchemes default Gauss linear
 * corrected; snGradSchemes default corrected; div((nuEff*dev(T(grad(U))))) Gauss linear;`).  Vectors [n][3], tensors
 * [n][9] (xx xy xz yx ...), symmTensors [n][6] (xx xy xz yy yz zz); patch-face arrays concatenated in
 * ldu_fv_boundary order; every pointer host or device; results bit-identical to the reference's loops.
 *
 * surfaceInterpolation::makeNonOrthDeltaCoeffs / makeNonOrthCorrectionVectors, internal faces
 * (surfaceInterpolation.C:289-305, :346-352): 1/max(unitArea & delta, 0.05*mag(delta)); unitArea -
 * delta*nonOrthDeltaCoeffs.  magSf may be NULL (= mag(Sf) + VSMALL, fvMeshGeometry.C:101-114); outputs may be NULL. */
int ldu_mesh_nonorth_factors(ldu_ctx* ctx, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                             const int32_t* neighbour, const double* faceAreas, const double* magSf,
                             const double* cellCentres, double* nonOrthDeltaCoeffs, double* nonOrthCorrectionVectors);
/* the faces of one patch (surfaceInterpolation.C:307-313, :362-390): patchDelta = fvPatch::delta(); correction
 * vectors are zero unless the patch is coupled */
int ldu_mesh_patch_nonorth_factors(ldu_ctx* ctx, int32_t nPatchFaces, const double* patchSf, const double* patchMagSf,
                                   const double* patchDelta, int32_t coupled, double* nonOrthDeltaCoeffs,
                                   double* nonOrthCorrectionVectors);
/* vec & linear.interpolate(field) on the internal faces, field = cell vectors (nComp 3 -> scalar per face) or cell
 * tensors (nComp 9 -> vector per face).  With vec = nonOrthCorrectionVectors and field = grad(vf) this is
 * correctedSnGrad<Type>::correction (correctedSnGrad.C:44-107, correctedSnGrads.C); with vec = Sf the face field of
 * gaussDivScheme::fvcDiv (gaussDivScheme.C:60-63); with vec = SfGammaCorr gammaSnGradCorr (gaussLaplacianScheme.C:92-128) */
int ldu_fv_interpolateDot(ldu_addr* a, int32_t nComp, const double* vec, const double* weights, const double* field,
                          double* out);
/* the same inner product for face values already at hand (patch faces): out = vec & field per face */
int ldu_fv_faceDot(ldu_ctx* ctx, int32_t nFaces, int32_t nComp, const double* vec, const double* field, double* out);
/* out = scale*field (accumulate 0) or out += scale*field (accumulate 1), scale per face: gammaMagSf*correction
 * (gaussLaplacianSchemes.C:64-66), tfaceFluxCorrection += SfGammaSn*correction (gaussLaplacianScheme.C:182-185) */
int ldu_fv_faceScale(ldu_ctx* ctx, int32_t nFaces, int32_t nComp, const double* scale, const double* field,
                     int32_t accumulate, double* out);
/* snGradScheme::snGrad(vf) of a corrected scheme (snGradScheme.C:168-186): nonOrthDeltaCoeffs*(vf[N]-vf[P]) +
 * correction (NULL: none) */
int ldu_fvc_correctedSnGrad(ldu_addr* a, int32_t nComp, const double* nonOrthDeltaCoeffs, const double* vf,
                            const double* correction, double* ssf);
/* surfaceInterpolationScheme::interpolate on the patch faces (surfaceInterpolationScheme.C:298-314): coupled faces
 * get w*vf[faceCells] + (1-w)*patchNeighbourField, the others patchValues (NULL: left as they are in out) */
int ldu_fv_interpolateBoundary(ldu_fv_boundary* b, int32_t nComp, const double* patchWeights, const double* vf,
                               const double* patchNeighbourField, const double* patchValues, double* out);
/* gaussGrad::correctBoundaryConditions (gaussGrad.C:144-170) on the ordinary patches: boundaryGrad = g +
 * n*(snGrad - (n & g)), g = grad[faceCells]; nComp 1 (grad [nCells][3]) or 3 ([nCells][9]).  Coupled faces of
 * boundaryGrad are not touched (ldu_fv_interpolateBoundary fills them). */
int ldu_fvc_gaussGradBoundary(ldu_fv_boundary* b, int32_t nComp, const double* patchNf, const double* grad,
                              const double* patchSnGrad, double* boundaryGrad);
/* fvc::surfaceIntegrate (= fvc::div of a face field) with its patch faces (fvcSurfaceIntegrate.C:43-76); nComp 1 or 3;
 * b / boundarySsf may be NULL.  Together with ldu_fv_interpolateDot(Sf, ...) this is gaussDivScheme::fvcDiv. */
int ldu_fvc_surfaceIntegrateFull(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* ssf,
                                 const double* boundarySsf, const double* V, double* out);
/* the explicit non-orthogonal source of fvm::laplacian: source -= V*fvc::div(faceFluxCorrection)
 * (gaussLaplacianSchemes.C:74-88, gaussLaplacianScheme.C:187) */
int ldu_fvm_sourceMinusVDiv(ldu_addr* a, ldu_fv_boundary* b, int32_t nComp, const double* faceFluxCorrection,
                            const double* boundaryFaceFluxCorrection, const double* V, double* source);
/* tensor diffusivity (gaussLaplacianScheme.C:165-173): Sn = Sf/magSf; SfGammaSn = (Sf & gamma) & Sn (feeds
 * ldu_fvm_laplacian in place of gammaMagSf); SfGammaCorr = (Sf & gamma) - SfGammaSn*Sn.  nGammaCmpt 6 | 9. */
int ldu_fv_tensorGammaFactors(ldu_ctx* ctx, int32_t nFaces, int32_t nGammaCmpt, const double* Sf, const double* magSf,
                              const double* gamma, double* SfGammaSn, double* SfGammaCorr);

/* ---- coupled solvers: LduMatrix<Type, scalar, scalar> (src/OpenFOAM/matrices/LduMatrix) -------------
 * `type coupled;` in fvSolution (fvMatrixSolve.C:83-85, solveCoupled :222-277) solves every component
 * of a vector/tensor field at once on the scalar coefficients of the same ldu_matrix.  Fields are
 * Field<Type> images: nCells x nCmpt doubles, components interleaved, original cell order
 * (nCmpt = pTraits<Type>::nComponents: 1, 3, 6 or 9).  The matrix source is passed with the call.
 * Selection follows LduMatrixSolver.C:33-111 / LduMatrixPreconditioner.C:33-103 / LduMatrixSmoother.C:
 * a matrix without faces is solved by DiagonalSolver whatever the name says; PCICG is registered for
 * symmetric matrices only, PBiCCCG / PBiCICG for asymmetric ones, SmoothSolver for both
 * (Solvers/lduSolvers.C:33-50); DILU is an asymmetric-matrix preconditioner only
 * (Preconditioners/lduPreconditioners.C:41-42) - a name missing from the table is an error (-16), as
 * the reference's FatalIOError. */
enum { LDU_CSOLVER_PCICG = 0, LDU_CSOLVER_PBICCCG = 1, LDU_CSOLVER_PBICICG = 2, LDU_CSOLVER_SMOOTHSOLVER = 3,
       LDU_CSOLVER_DIAGONAL = 4 };
enum { LDU_CPRE_NONE = 0, LDU_CPRE_DIAGONAL = 1, LDU_CPRE_DILU = 2 };
enum { LDU_CSM_GAUSSSEIDEL = 0 };
#define LDU_MAX_CMPT 9

/* LduMatrix::solver controls (LduMatrixSolver.C:120-149): maxIter 1000, tolerance 1e-6*one, relTol zero;
 * SmoothSolver.C:53-57 nSweeps. */
typedef struct ldu_coupled_controls {
    int32_t solver, preconditioner, smoother;
    int32_t nCmpt;
    int32_t maxIter, nSweeps;
    double tolerance[LDU_MAX_CMPT], relTol[LDU_MAX_CMPT];
    /* PBiCCCG's gSumProd is the Type's double inner product `&&` per cell: weight of component k =
     * (unit_k && unit_k): all 1 for vector / tensor, 1 2 2 1 2 1 for symmTensor (SymmTensorI.H:212-220),
     * 3 for sphericalTensor (SphericalTensorI.H:130-133).  ldu_coupled_default_controls sets them from nCmpt
     * (6 = symmTensor; 1 = scalar). */
    double innerProductWeights[LDU_MAX_CMPT];
} ldu_coupled_controls;

/* SolverPerformance<Type> (LduMatrix/SolverPerformance.H) */
typedef struct ldu_coupled_perf {
    double initialResidual[LDU_MAX_CMPT], finalResidual[LDU_MAX_CMPT], normFactor[LDU_MAX_CMPT];
    int32_t singular[LDU_MAX_CMPT];
    int32_t nIterations, converged;
    double solveSeconds;
} ldu_coupled_perf;

void ldu_coupled_default_controls(ldu_coupled_controls* c, int32_t nCmpt);
/* LduMatrix::solver::solve(psi) (LduMatrix.H:238): PCICG.C:50-184, PBiCCCG.C:50-192, PBiCICG.C:50-197,
 * SmoothSolver.C:61-151, DiagonalSolver.C:56-76 */
int ldu_coupled_solve(ldu_matrix* m, const ldu_coupled_controls* controls, double* psi, const double* source,
                      ldu_coupled_perf* perf);
/* LduMatrix::Amul / Tmul / residual (LduMatrixATmul.C:66-114, :117-165, :218-276) */
int ldu_coupled_amul(ldu_matrix* m, int32_t nCmpt, double* Apsi, const double* psi, int32_t transpose);
int ldu_coupled_residual(ldu_matrix* m, int32_t nCmpt, double* rA, const double* psi, const double* source);
/* LduMatrix::preconditioner::precondition / preconditionT (TDILUPreconditioner.C:82-125, :128-176;
 * DiagonalPreconditioner.C:64-80; NoPreconditioner.C:49-56) */
int ldu_coupled_precondition(ldu_matrix* m, int32_t preconditioner, int32_t nCmpt, double* wA, const double* rA,
                             int32_t transpose);
/* LduMatrix::smoother::smooth (TGaussSeidelSmoother.C:63-153) */
int ldu_coupled_smooth(ldu_matrix* m, int32_t smoother, int32_t nCmpt, double* psi, const double* source,
                       int32_t nSweeps);

/* ---- mesh side: polyMesh arrays -> geometric fields, renumbering ----------------
 * Faces as a CSR of point labels (faceStart[nFaces+1], facePoints), all faces (internal first, then the
 * patches, as in constant/polyMesh/faces); owner for every face, neighbour for the internal ones.
 * primitiveMesh::makeFaceCentresAndAreas (primitiveMeshFaceCentresAndAreas.C:73-131) and
 * makeCellCentresAndVols (primitiveMeshCellCentresAndVols.C:72-147); arrays host or device. */
int ldu_mesh_geometry(ldu_ctx* ctx, int32_t nPoints, const double* points, int32_t nFaces, const int32_t* faceStart,
                      const int32_t* facePoints, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                      const int32_t* neighbour, double* faceCentres /* nFaces*3 */, double* faceAreas /* nFaces*3 */,
                      double* cellCentres /* nCells*3 */, double* cellVolumes /* nCells */);
/* surfaceInterpolation::makeWeights / makeDeltaCoeffs, internal faces (surfaceInterpolation.C:163-185, :227-231)
 * and fvMesh::magSf; any output may be NULL */
int ldu_mesh_interpolation_factors(ldu_ctx* ctx, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                                   const int32_t* neighbour, const double* faceCentres, const double* faceAreas,
                                   const double* cellCentres, double* weights, double* deltaCoeffs, double* magSf);
/* Foam::bandCompression (meshes/bandCompression/bandCompression.C:43-146) on the cell-cell addressing of the
 * internal faces: newOrder[i] = old label of the cell that becomes cell i (what renumberMesh applies) */
int ldu_band_compression(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                         int32_t* newOrder);
/* A numbering for manualRenumber (src/renumber/renumberMethods/manualRenumber/manualRenumber.C:60-136 reads exactly this
 * newToOld list): `order` (NULL = identity; normally ldu_band_compression's) cut into tiles of tileSize consecutive cells, the
 * tiles re-ordered by a hash of (seed, tile index), the order inside a tile kept.  Keeps bandCompression's sweep inside a tile
 * and bounds the GaussSeidel dependency chains of every GAMG level by a handful of tiles (DESIGN "Numbering"); host code. */
int ldu_tile_shuffle(int32_t nCells, const int32_t* order, int32_t tileSize, uint64_t seed, int32_t* newOrder);
/* nParts compact sub-domains of nearly equal size (breadth-first blobs in the order of the numbering; host code): the cut
 * ldu_addr_set_subdomains' callers use when the case brings no decomposition of its own */
int ldu_partition_blobs(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr, int32_t nParts,
                        int32_t* part);
/* the block engine's own cut of a level whose equal-size blobs do not fit into a workgroup's LDS (csrc/ldu_blocks.hip): breadth-
 * first blobs of equal FOOTPRINT - a blob grows until its cells plus the distinct cells outside it that touch it reach
 * slotTarget -, as many as that takes (*nParts; more than maxParts: error).  Host code; exported for tests and tools. */
int ldu_partition_blobs_footprint(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                                  int64_t slotTarget, int32_t maxParts, int32_t* part, int32_t* nParts);
/* the matrix addressing under such a renumbering, back in upper-triangular order (lduAddressing.C:92-126 needs
 * it): faceMap[newFace] = old face, flip[newFace] = 1 when lower/upper of that face swap (may be NULL) */
int ldu_renumber_addressing(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                            const int32_t* newOrder, int32_t* newLower, int32_t* newUpper, int32_t* faceMap,
                            uint8_t* flip);

/* Host-only views of the halo wire protocol (no device, no communicator; tests/test_gloo_2rank.py):
 * ldu_comm_paired_patch: index, on the neighbour rank, of the patch that pairs with my patch p - the k-th patch of the
 *   neighbour towards me, k = ordinal of p among my patches towards that neighbour (processorFvPatch pairs by
 *   myProcNo / neighbProcNo the same way); -1 = none.
 * ldu_comm_exchange_order: the patches for which one operator application issues an ncclSend + ncclRecv, in issue
 *   order (patches with faces that are not cyclic: nbrPatch < 0); returns their number. */
int ldu_comm_paired_patch(int32_t nMine, const int32_t* mineNbrRank, int32_t p, int32_t nTheirs,
                          const int32_t* theirsNbrRank, int32_t me);
int ldu_comm_exchange_order(int32_t nPatches, const int32_t* nFaces, const int32_t* nbrPatch, int32_t* order);

/* Host-only plan statistics of an addressing (no device needed): dependency levels of the lower-triangular DAG
 * (what bounds every sequential sweep of the reference: GaussSeidelSmoother.C:147-176, DICPreconditioner.C:71-122)
 * and the cluster partition of the cluster sweep engine.  out[0] dependency levels, [1] clusters, [2] cluster levels,
 * [3] sum and [4] maximum of the clusters' internal steps, [5]/[6] most lower/upper neighbours of a row, [7]/[8] rows
 * with more than 6/12 lower or upper neighbours, [9] faces between clusters, [10]/[11] widest dependency / cluster
 * level.  cellLevel[nCells], cellCluster[nCells], clusterLevel[>= clusters] may be NULL. */
int ldu_debug_dag_stats(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                        int32_t maxCellsPerCluster, int64_t out[16], int32_t* cellLevel, int32_t* cellCluster,
                        int32_t* clusterLevel);

#ifdef __cplusplus
}
#endif
#endif
