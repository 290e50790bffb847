/*---------------------------------------------------------------------------*\
  hipLduSolvers - OpenFOAM-2.2.x plugin that puts the MI355X lduMatrix hot path
  (libldugpu.so, include/ldugpu.h) behind OpenFOAM's own run-time selection:

      system/controlDict:   libs ("libhipLduSolvers.so");

  After the library is loaded, `solver PCG; / PBiCG; / GAMG; / smoothSolver;`
  in system/fvSolution resolve to the classes below (registered with
  addRemovable...ConstructorToTable, which REPLACES the stock table entry:
  runTimeSelectionTables.H:102-133); the same classes are also reachable as
  hipPCG / hipPBiCG / hipGAMG / hipSmoothSolver.  No solver application changes.

  This file is the ONLY code that sees OpenFOAM types: it marshals the raw
  arrays the reference hands to its solvers into the C ABI.
  Built against the reference headers by plugin/build_plugin.sh.
\*---------------------------------------------------------------------------*/

#include "lduMatrix.H"
#include "LduMatrix.H"
#include "fieldTypes.H"
#include "processorLduInterface.H"
#include "cyclicLduInterface.H"
#include "cyclicLduInterfaceField.H"
#include "processorLduInterfaceField.H"
#include "addToRunTimeSelectionTable.H"
#include "Pstream.H"
#include "Switch.H"
#include "labelList.H"
#include "polyMesh.H"

#include <cstring>
#include <cstdlib>

#include "ldugpu.h"

#include <map>
#include <string>
#include <cstdio>
#include <vector>
#include <stdint.h>

namespace Foam
{

// ------------------------------------------------------------------ device registry

// One entry per lduAddressing the solvers have seen.  The key is the address of the addressing object; an
// entry is only reused when its fingerprint still matches: the two address arrays (pointer, size, a strided
// sample of their contents) and every coupled patch's faceCells (size + sample).  The sample cannot see a local
// topology change that keeps sizes, base pointers and the sampled entries: on a mesh that says it is changing
// (polyMesh::changing(): points moved or topology changed this time step) the WHOLE lists are hashed and compared
// with the hash taken when the entry was built, and the geometric agglomeration weights are taken again when the
// points move (polyMesh::moving()).  An lduAddressing freed and
// re-allocated at the same address (GAMG levels without cacheAgglomeration, a topology change that keeps the
// cell and face counts) therefore rebuilds its device plan instead of silently reusing a stale one.
// Entries are evicted least-recently-used beyond hipMaxEntries_ and all freed when the library is unloaded.
struct hipLduEntry
{
    ldu_addr* addr;
    ldu_matrix* mat;
    label nCells, nFaces;
    uint64_t fingerprint;
    uint64_t fullFingerprint;     // every entry of every list (taken at construction, re-checked on changing meshes)
    uint64_t lastUse;
    bool weightsSet;
};

static ldu_ctx* hipCtx_ = NULL;
static std::map<const lduAddressing*, hipLduEntry> hipEntries_;
static uint64_t hipUseClock_ = 0;
static const size_t hipMaxEntries_ = 64;

static void hipCheck(int rc, const char* where)
{
    if (rc)
    {
        FatalErrorIn(where) << "libldugpu: " << ldu_last_error() << exit(FatalError);
    }
}

static inline void hipMix(uint64_t& h, uint64_t v)
{
    h ^= v + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
}

static void hipMixList(uint64_t& h, const labelUList& l, const bool full)
{
    hipMix(h, uint64_t(l.size()));
    hipMix(h, uint64_t(reinterpret_cast<uintptr_t>(l.begin())));
    const label n = l.size();
    const label step = (!full && n > 4096) ? n/4096 : 1;
    for (label i = 0; i < n; i += step) hipMix(h, uint64_t(l[i]));
    if (n) hipMix(h, uint64_t(l[n - 1]));
}

template<class InterfaceList>
static uint64_t hipFingerprint(const lduAddressing& la, const InterfaceList& interfaces, const bool full = false)
{
    uint64_t h = 0x243F6A8885A308D3ULL;
    hipMix(h, uint64_t(la.size()));
    hipMixList(h, la.lowerAddr(), full);
    hipMixList(h, la.upperAddr(), full);
    forAll(interfaces, patchi)
    {
        hipMix(h, interfaces.set(patchi) ? 1 : 0);
        if (interfaces.set(patchi)) hipMixList(h, la.patchAddr(patchi), full);
    }
    return h;
}

// the polyMesh behind a matrix (fvMesh is one; the GAMG levels' lduPrimitiveMesh is not)
static const polyMesh* hipPolyMesh(const lduMesh& mesh)
{
    const polyMesh* pm = dynamic_cast<const polyMesh*>(&mesh);
    if (!pm) pm = dynamic_cast<const polyMesh*>(&mesh.thisDb());
    return pm;
}

static void hipFreeEntry(hipLduEntry& e)
{
    ldu_matrix_destroy(e.mat);
    ldu_addr_destroy(e.addr);
}

static void hipEvictOldest()
{
    std::map<const lduAddressing*, hipLduEntry>::iterator old = hipEntries_.begin();
    for (std::map<const lduAddressing*, hipLduEntry>::iterator i = hipEntries_.begin(); i != hipEntries_.end(); ++i)
    {
        if (i->second.lastUse < old->second.lastUse) old = i;
    }
    hipFreeEntry(old->second);
    hipEntries_.erase(old);
}

// device memory is returned when the library is unloaded (dlclose / exit).  Not in a parallel run: static destructors
// run after Pstream::exit() / MPI_Finalize and possibly during the HIP / RCCL runtime's own teardown, where
// ncclCommDestroy and stream synchronisation can hang or crash - the process is ending, the driver reclaims the memory.
static bool hipParallelRun_ = false;
struct hipLduRegistryCleaner
{
    ~hipLduRegistryCleaner()
    {
        if (hipParallelRun_) return;
        for (std::map<const lduAddressing*, hipLduEntry>::iterator i = hipEntries_.begin(); i != hipEntries_.end(); ++i)
        {
            hipFreeEntry(i->second);
        }
        hipEntries_.clear();
        if (hipCtx_) { ldu_ctx_destroy(hipCtx_); hipCtx_ = NULL; }
    }
};
static hipLduRegistryCleaner hipLduRegistryCleaner_;

// rank -> GPU: the rank's position within its node (what the MPI launchers export) modulo the devices this
// process can see (HIP_VISIBLE_DEVICES respected); without a launcher hint, the global rank modulo the count
static int hipDeviceForRank()
{
    const int nDev = ldu_device_count();
    if (nDev <= 0)
    {
        FatalErrorIn("hipDeviceForRank()") << "no HIP device visible to this process" << exit(FatalError);
    }
    if (!Pstream::parRun()) return 0;
    const char* vars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID",
                          "SLURM_LOCALID", "PMI_LOCAL_RANK"};
    for (unsigned i = 0; i < sizeof(vars)/sizeof(vars[0]); i++)
    {
        const char* v = getenv(vars[i]);
        if (v && *v) return atoi(v) % nDev;
    }
    return Pstream::myProcNo() % nDev;
}

// Out-of-band exchange of the peer-store backend's set-up messages (ldu_ctx_comm_init_peer, include/ldugpu.h) over the
// application's own Pstream - where the reference itself sends the restrict maps of the coarse processor interfaces
// (processorGAMGInterface.C:137-154) and reduces continueAgglomerating (GAMGAgglomeration.C:53-62).  Pairwise and
// non-blocking: every rank of a pair posts the receive, then the send, then waits for both (UIPstream::read /
// UOPstream::write with Pstream::nonBlocking, UPstream::waitRequests: lduMatrixUpdateMatrixInterfaces.C:127-160 uses the
// same calls for the patch fields).
static int hipOobExchange
(
    void*, int32_t nPeers, const int32_t* peers, const void* const* sendBufs, const int64_t* sendBytes,
    void* const* recvBufs, const int64_t* recvBytes
)
{
    const label startReq = UPstream::nRequests();
    for (int32_t i = 0; i < nPeers; i++)
    {
        if (recvBytes[i] > 0)
        {
            UIPstream::read
            (
                Pstream::nonBlocking, peers[i], reinterpret_cast<char*>(recvBufs[i]), std::streamsize(recvBytes[i])
            );
        }
    }
    for (int32_t i = 0; i < nPeers; i++)
    {
        if (sendBytes[i] > 0)
        {
            UOPstream::write
            (
                Pstream::nonBlocking, peers[i], reinterpret_cast<const char*>(sendBufs[i]), std::streamsize(sendBytes[i])
            );
        }
    }
    UPstream::waitRequests(startReq);
    return 0;
}

static ldu_ctx* hipContext()
{
    if (!hipCtx_)
    {
        // one rank per GPU
        const int dev = hipDeviceForRank();
        hipCheck(ldu_ctx_create(&hipCtx_, dev), "hipContext()");
        if (Pstream::parRun())
        {
            hipParallelRun_ = true;
            // lduMatrixUpdateMatrixInterfaces.C:30-266: `blocking` and `nonBlocking` update the coupled patches in patch
            // order - the order this library applies them in, whatever moves the data; `scheduled` walks the patch
            // schedule instead, which changes the order in which the contributions of SEVERAL coupled patches reach
            // one cell (last-bit differences there).  Said once, not an error: the exchange itself is ours (RCCL).
            if (Pstream::defaultCommsType == Pstream::scheduled)
            {
                WarningIn("hipContext()")
                    << "commsType scheduled: libhipLduSolvers updates coupled patches in patch order (the blocking / "
                    << "nonBlocking order of lduMatrix::updateMatrixInterfaces); cells on several coupled patches may "
                    << "differ from the scheduled order in the last bits" << endl;
            }
            // RCCL communicator bootstrapped over the existing Pstream (replaces MPI on the hot path)
            labelList id(128/sizeof(label), 0);      // 128-byte ncclUniqueId as labels
            uint8_t raw[128];
            if (Pstream::master())
            {
                hipCheck(ldu_comm_unique_id(raw), "hipContext()");
                std::memcpy(id.begin(), raw, 128);
            }
            Pstream::scatter(id);
            std::memcpy(raw, id.begin(), 128);
            hipCheck
            (
                ldu_ctx_comm_init(hipCtx_, Pstream::myProcNo(), Pstream::nProcs(), raw),
                "hipContext()"
            );
            // Intra-node runs (all ranks on the GPUs of one xGMI-connected node, <= 16 ranks): the peer-store windows next
            // to the RCCL communicator.  RCCL stays the carrier unless LDU_HALO=p2p / LDU_REDUCE=p2p say otherwise
            // (ldugpu.h); LDU_PEER=0 skips the windows.  Ranks on different hosts cannot map each other's memory: the
            // host names are compared first and the windows are only opened when they are all equal.
            const char* pe = getenv("LDU_PEER");
            if ((!pe || atoi(pe)) && Pstream::nProcs() <= 16)
            {
                List<word> hosts(Pstream::nProcs());
                hosts[Pstream::myProcNo()] = hostName();
                Pstream::gatherList(hosts);
                Pstream::scatterList(hosts);
                bool oneNode = true;
                forAll(hosts, i) oneNode = oneNode && hosts[i] == hosts[0];
                if (oneNode)
                {
                    hipCheck
                    (
                        ldu_ctx_comm_init_peer(hipCtx_, Pstream::myProcNo(), Pstream::nProcs(), &hipOobExchange, NULL),
                        "hipContext()"
                    );
                }
            }
        }
    }
    return hipCtx_;
}

// the process's one device context, for the scheme plug-in (hipFvSchemes.C) that shares it
ldu_ctx* hipLduSharedContext()
{
    return hipContext();
}

// A coupled patch with a transformation (rotational cyclic, processorCyclic with rotation): the interface FIELD multiplies the
// neighbour values by pow(diag(forwardT).component(cmpt), rank()) before the coefficients
// (cyclicLduInterfaceField.C:45-63, processorLduInterfaceField.C:45-63).  For a field of rank 0 - p, k, epsilon, T: every
// scalar transport equation - that factor is pow(x, 0) = 1 and the product is the value itself, bit for bit: the patch is an
// ordinary coupled patch for this solve.  For a component of a vector / tensor field the factor is a rotation-dependent number:
// not implemented, refused.  The check runs at EVERY solve: the device image of an addressing is shared by all fields on it.
static inline bool hipTransformIsIdentity(const lduInterfaceField& f)
{
    const cyclicLduInterfaceField* c = dynamic_cast<const cyclicLduInterfaceField*>(&f);
    if (c) return !c->doTransform() || c->rank() == 0;
    const processorLduInterfaceField* p = dynamic_cast<const processorLduInterfaceField*>(&f);
    if (p) return !p->doTransform() || p->rank() == 0;
    return false;
}
// LduInterfaceField<Type> (the `type coupled;` family) transforms the Type itself: a transformation is never the identity there
template<class Type>
static inline bool hipTransformIsIdentity(const LduInterfaceField<Type>&) { return false; }

template<class InterfaceList>
static void hipCheckTransforms(const InterfaceList& interfaces)
{
    forAll(interfaces, patchi)
    {
        if (!interfaces.set(patchi)) continue;
        const lduInterface& li = interfaces[patchi].interface();
        const processorLduInterface* pp = dynamic_cast<const processorLduInterface*>(&li);
        const cyclicLduInterface* cp = dynamic_cast<const cyclicLduInterface*>(&li);
        const bool transformed = (pp && pp->forwardT().size()) || (cp && cp->forwardT().size());
        if (transformed && !hipTransformIsIdentity(interfaces[patchi]))
        {
            FatalErrorIn("hipLookup")
                << (pp ? "processor" : "cyclic") << " patch " << patchi << " carries a transformation (rotation) and the field "
                << "solved for is not of rank 0: only scalar fields (whose coupling the transformation leaves unchanged, "
                << "cyclicLduInterfaceField.C:45-63) are supported across transformed patches on the GPU path"
                << exit(FatalError);
        }
    }
}

// Device image of (lduAddressing, coupled patches), built once per addressing like the
// reference's lazily built losort/ownerStart.
template<class InterfaceList>
static hipLduEntry& hipLookupAddr
(
    const lduAddressing& la,
    const InterfaceList& interfaces,
    const polyMesh* pm = NULL
)
{
    hipCheckTransforms(interfaces);
    std::map<const lduAddressing*, hipLduEntry>::iterator it = hipEntries_.find(&la);
    const label nCells = la.size();
    const label nFaces = la.lowerAddr().size();
    const uint64_t fp = hipFingerprint(la, interfaces);
    const bool changing = pm && pm->changing();
    if (it != hipEntries_.end() && it->second.fingerprint == fp && changing)
    {
        // a mesh in motion / with topology changes: the sample is not enough
        if (hipFingerprint(la, interfaces, true) != it->second.fullFingerprint) it->second.fingerprint = ~fp;
        if (pm->moving()) it->second.weightsSet = false;   // face areas changed: faceAreaPair weights again
    }
    if (it != hipEntries_.end() && it->second.fingerprint != fp)
    {
        hipFreeEntry(it->second);
        hipEntries_.erase(it);
        it = hipEntries_.end();
    }
    if (it == hipEntries_.end())
    {
        while (hipEntries_.size() >= hipMaxEntries_) hipEvictOldest();
        hipLduEntry e;
        e.nCells = nCells;
        e.nFaces = nFaces;
        e.fingerprint = fp;
        e.fullFingerprint = hipFingerprint(la, interfaces, true);
        e.lastUse = 0;
        e.weightsSet = false;
        hipCheck
        (
            ldu_addr_create
            (
                hipContext(), &e.addr, nCells, nFaces,
                la.lowerAddr().begin(), la.upperAddr().begin()
            ),
            "hipLookup"
        );
        bool any = false;
        // position of every coupled patch in the library's patch list (= order of addition)
        labelList libIndex(interfaces.size(), -1);
        {
            label k = 0;
            forAll(interfaces, patchi) if (interfaces.set(patchi)) libIndex[patchi] = k++;
        }
        forAll(interfaces, patchi)
        {
            if (interfaces.set(patchi))
            {
                const lduInterface& li = interfaces[patchi].interface();
                const processorLduInterface* pp = dynamic_cast<const processorLduInterface*>(&li);
                const cyclicLduInterface* cp = dynamic_cast<const cyclicLduInterface*>(&li);
                const labelUList& fc = la.patchAddr(patchi);
                if (pp)
                {
                    // (a transformation on the patch: hipCheckTransforms above - identity for the rank-0 field of this solve)
                    if (pp->forwardT().size() && getenv("LDU_VERBOSE"))
                        Info<< "[hipLduSolvers] processor patch " << patchi << " carries a transformation: identity for a field of "
                            << "rank 0, coupled as an ordinary processor patch" << endl;
                    hipCheck(ldu_addr_add_patch(e.addr, fc.size(), fc.begin(), pp->neighbProcNo()), "hipLookup");
                }
                else if (cp)
                {
                    if (cp->forwardT().size() && getenv("LDU_VERBOSE"))
                        Info<< "[hipLduSolvers] cyclic patch " << patchi << " carries a transformation (rotational cyclic): identity "
                            << "for a field of rank 0, coupled as an ordinary cyclic patch" << endl;
                    const label nb = cp->neighbPatchID();
                    if (nb < 0 || nb >= libIndex.size() || libIndex[nb] < 0)
                    {
                        FatalErrorIn("hipLookup") << "cyclic patch " << patchi << ": neighbour patch " << nb
                            << " is not a coupled interface of this matrix" << exit(FatalError);
                    }
                    hipCheck(ldu_addr_add_cyclic_patch(e.addr, fc.size(), fc.begin(), libIndex[nb]), "hipLookup");
                }
                else
                {
                    FatalErrorIn("hipLookup")
                        << "only processor and cyclic coupled patches are supported on the GPU path, patch "
                        << patchi << " is " << li.type() << exit(FatalError);
                }
                any = true;
            }
        }
        if (any) hipCheck(ldu_addr_finalize(e.addr), "hipLookup");
        hipCheck(ldu_matrix_create(e.addr, &e.mat), "hipLookup");
        it = hipEntries_.insert(std::make_pair(&la, e)).first;
    }
    it->second.lastUse = ++hipUseClock_;
    return it->second;
}

static hipLduEntry& hipLookup
(
    const lduMatrix& matrix,
    const lduInterfaceFieldPtrsList& interfaces
)
{
    return hipLookupAddr(matrix.lduAddr(), interfaces, hipPolyMesh(matrix.mesh()));
}

// Coefficients are re-read every solve (fvScalarMatrix.C:152-174 changes diag around the call).
static void hipSetCoeffs
(
    hipLduEntry& e,
    const lduMatrix& matrix,
    const FieldField<Field, scalar>& bouCoeffs,
    const FieldField<Field, scalar>& intCoeffs,
    const lduInterfaceFieldPtrsList& interfaces
)
{
    hipCheck
    (
        ldu_matrix_set_coeffs
        (
            e.mat, matrix.diag().begin(), matrix.upper().begin(),
            matrix.asymmetric() ? matrix.lower().begin() : NULL
        ),
        "hipSetCoeffs"
    );
    label k = 0;
    forAll(interfaces, patchi)
    {
        if (interfaces.set(patchi))
        {
            hipCheck
            (
                ldu_matrix_set_patch_coeffs(e.mat, k++, bouCoeffs[patchi].begin(), intCoeffs[patchi].begin()),
                "hipSetCoeffs"
            );
        }
    }
}

static int hipPreconditionerKind(const word& n)
{
    if (n == "none") return LDU_PRE_NONE;
    if (n == "diagonal") return LDU_PRE_DIAGONAL;
    if (n == "DIC" || n == "hipDIC") return LDU_PRE_DIC;
    if (n == "FDIC") return LDU_PRE_FDIC;
    if (n == "DILU" || n == "hipDILU") return LDU_PRE_DILU;
    if (n == "GAMG") return LDU_PRE_GAMG;
    FatalErrorIn("hipPreconditionerKind") << "preconditioner " << n << " has no GPU implementation"
        << exit(FatalError);
    return -1;
}

static int hipSmootherKind(const word& n)
{
    if (n == "GaussSeidel" || n == "hipGaussSeidel") return LDU_SM_GAUSSSEIDEL;
    if (n == "symGaussSeidel") return LDU_SM_SYMGAUSSSEIDEL;
    if (n == "nonBlockingGaussSeidel") return LDU_SM_NONBLOCKINGGAUSSSEIDEL;
    if (n == "DIC") return LDU_SM_DIC;
    if (n == "DILU") return LDU_SM_DILU;
    if (n == "FDIC") return LDU_SM_FDIC;
    if (n == "DICGaussSeidel") return LDU_SM_DICGAUSSSEIDEL;
    if (n == "DILUGaussSeidel") return LDU_SM_DILUGAUSSSEIDEL;
    FatalErrorIn("hipSmootherKind") << "smoother " << n << " has no GPU implementation"
        << exit(FatalError);
    return -1;
}

// The keys the reference reads (lduMatrixSolver.C:164-169, smoothSolver.C:73, GAMGSolver.C:157-181,
// GAMGAgglomeration.C:79, pairGAMGAgglomeration.C:45, GAMGPreconditioner.C:77); preconditioner /
// smoother may be a word or a sub-dictionary (lduMatrixPreconditioner.C:70-81).
static void hipReadControls(const dictionary& dict, int solverKind, ldu_controls& c)
{
    ldu_default_controls(&c);
    c.solver = solverKind;
    c.maxIter = dict.lookupOrDefault<label>("maxIter", 1000);
    c.tolerance = dict.lookupOrDefault<scalar>("tolerance", 1e-6);
    c.relTol = dict.lookupOrDefault<scalar>("relTol", 0);
    c.nSweeps = dict.lookupOrDefault<label>("nSweeps", 1);
    const dictionary* gd = &dict;
    if (dict.found("preconditioner"))
    {
        c.preconditioner = hipPreconditionerKind(lduMatrix::preconditioner::getName(dict));
        if (dict.isDict("preconditioner")) gd = &dict.subDict("preconditioner");
    }
    if (gd->found("smoother"))
    {
        c.smoother = hipSmootherKind(lduMatrix::smoother::getName(*gd));
    }
    Switch sw(false);
    if (gd->readIfPresent("cacheAgglomeration", sw)) c.cacheAgglomeration = sw;
    label l;
    if (gd->readIfPresent("nPreSweeps", l)) c.nPreSweeps = l;
    if (gd->readIfPresent("preSweepsLevelMultiplier", l)) c.preSweepsLevelMultiplier = l;
    if (gd->readIfPresent("maxPreSweeps", l)) c.maxPreSweeps = l;
    if (gd->readIfPresent("nPostSweeps", l)) c.nPostSweeps = l;
    if (gd->readIfPresent("postSweepsLevelMultiplier", l)) c.postSweepsLevelMultiplier = l;
    if (gd->readIfPresent("maxPostSweeps", l)) c.maxPostSweeps = l;
    if (gd->readIfPresent("nFinestSweeps", l)) c.nFinestSweeps = l;
    if (gd->readIfPresent("interpolateCorrection", sw)) c.interpolateCorrection = sw;
    if (gd->readIfPresent("scaleCorrection", sw)) c.scaleCorrection = sw;
    if (gd->readIfPresent("directSolveCoarsest", sw)) c.directSolveCoarsest = sw;
    if (gd->readIfPresent("nCellsInCoarsestLevel", l)) c.nCellsInCoarsestLevel = l;
    if (gd->readIfPresent("mergeLevels", l)) c.mergeLevels = l;
    if (gd->readIfPresent("nVcycles", l)) c.nVcycles = l;
    if (gd->found("agglomerator"))
    {
        const word ag(gd->lookup("agglomerator"));
        if (ag == "algebraicPair") c.agglomerator = LDU_AGG_ALGEBRAICPAIR;
        else if (ag == "faceAreaPair") c.agglomerator = LDU_AGG_FACEAREAPAIR;
        else
        {
            FatalErrorIn("hipReadControls") << "agglomerator " << ag << " has no GPU implementation "
                << "(available: faceAreaPair, algebraicPair)" << exit(FatalError);
        }
    }
}


// Geometric agglomeration weights, obtained the way faceAreaPairGAMGAgglomeration.C:48-73 obtains them: from
// the mesh behind matrix.mesh().  The reference refCasts the lduMesh to fvMesh and uses Sf() / magSf(); the
// internal field of fvMesh::Sf() is the first nInternalFaces entries of primitiveMesh::faceAreas()
// (fvMeshGeometry.C:52-76: a sliced field over faceAreas()) and magSf = mag(Sf) + VSMALL (:101-114), so the
// polyMesh base of the same object (libOpenFOAM - the shim does not link libfiniteVolume) gives the same numbers.
// mag(cmptMultiply(Sf/sqrt(magSf), (1, 1.01, 1.02))) itself is evaluated on the device (ldu_addr_set_face_areas).
static void hipEnsureFaceWeights(hipLduEntry& e, const lduMatrix& matrix)
{
    if (e.weightsSet) return;
    const polyMesh* pm = hipPolyMesh(matrix.mesh());
    if (!pm || pm->nInternalFaces() != e.nFaces)
    {
        // the reference: refCast<const fvMesh>(mesh) fails the same way on anything that is not an fvMesh
        FatalErrorIn("hipEnsureFaceWeights")
            << "agglomerator faceAreaPair needs the face areas of the finite-volume mesh behind the matrix "
            << "(faceAreaPairGAMGAgglomeration.C:56); this matrix's lduMesh is " << (pm ? "a polyMesh whose internal "
               "faces do not match the matrix addressing" : "not a polyMesh") << ". Use agglomerator algebraicPair."
            << exit(FatalError);
    }
    const vectorField& Sf = pm->faceAreas();
    hipCheck(ldu_addr_set_face_areas(e.addr, reinterpret_cast<const double*>(Sf.begin())), "hipEnsureFaceWeights");
    e.weightsSet = true;
}


// LDU_DUMP_MATRIX="<field>:<n>:<file>": the n-th (1-based) matrix a solver of this plug-in is handed for <field> is written
// to <file> before it is solved - int64 {nCells, nFaces, symmetric, hasSf}, lowerAddr, upperAddr (int32), diag, upper,
// lower (asymmetric only), source, psi (doubles), Sf of the internal faces (3 doubles each, when the mesh is a polyMesh).
// Matrices without coupled interfaces only.  The test / probe side reads it back (tests/test_simplefoam_motorbike.py):
// a p-matrix of a real SIMPLE iteration on a real mesh as a stand-alone workload for the C ABI.
static void hipDumpMatrix
(
    const word& fieldName, const lduMatrix& matrix, const lduInterfaceFieldPtrsList& interfaces,
    const scalarField& psi, const scalarField& source
)
{
    static const char* spec = getenv("LDU_DUMP_MATRIX");
    if (!spec) return;
    static std::map<std::string, int> calls;
    const std::string s(spec);
    const size_t a = s.find(':'), b = s.find(':', a == std::string::npos ? a : a + 1);
    if (a == std::string::npos || b == std::string::npos) return;
    if (s.substr(0, a) != fieldName) return;
    if (++calls[fieldName] != atoi(s.substr(a + 1, b - a - 1).c_str())) return;
    forAll(interfaces, i)
    {
        if (interfaces.set(i))
        {
            WarningIn("hipDumpMatrix") << "matrix with coupled interfaces: not dumped" << endl;
            return;
        }
    }
    FILE* f = fopen(s.substr(b + 1).c_str(), "wb");
    if (!f) return;
    const lduAddressing& ad = matrix.lduAddr();
    const polyMesh* pm = hipPolyMesh(matrix.mesh());
    const label nF = ad.lowerAddr().size();
    const bool sf = pm && pm->nInternalFaces() == nF;
    long long head[4] = {psi.size(), nF, matrix.symmetric() || matrix.diagonal(), sf};
    fwrite(head, sizeof(long long), 4, f);
    std::vector<int> idx(nF);
    forAll(ad.lowerAddr(), i) idx[i] = ad.lowerAddr()[i];
    fwrite(idx.data(), sizeof(int), nF, f);
    forAll(ad.upperAddr(), i) idx[i] = ad.upperAddr()[i];
    fwrite(idx.data(), sizeof(int), nF, f);
    fwrite(matrix.diag().begin(), sizeof(double), psi.size(), f);
    if (matrix.hasUpper()) fwrite(matrix.upper().begin(), sizeof(double), nF, f);
    else { std::vector<double> z(nF, 0.0); fwrite(z.data(), sizeof(double), nF, f); }
    if (!head[2]) fwrite(matrix.lower().begin(), sizeof(double), nF, f);
    fwrite(source.begin(), sizeof(double), psi.size(), f);
    fwrite(psi.begin(), sizeof(double), psi.size(), f);
    if (sf) fwrite(pm->faceAreas().begin(), sizeof(double), 3*size_t(nF), f);
    fclose(f);
    Info<< "[hipLduSolvers] matrix of " << fieldName << " (call " << calls[fieldName] << ") written to "
        << s.substr(b + 1).c_str() << endl;
}


// ------------------------------------------------------------------ solvers

template<int SolverKind>
class hipLduSolver
:
    public lduMatrix::solver
{
public:

    static const word typeName;
    virtual const word& type() const { return typeName; }

    hipLduSolver
    (
        const word& fieldName,
        const lduMatrix& matrix,
        const FieldField<Field, scalar>& interfaceBouCoeffs,
        const FieldField<Field, scalar>& interfaceIntCoeffs,
        const lduInterfaceFieldPtrsList& interfaces,
        const dictionary& solverControls
    )
    :
        lduMatrix::solver
        (
            fieldName, matrix, interfaceBouCoeffs, interfaceIntCoeffs, interfaces, solverControls
        )
    {}

    virtual ~hipLduSolver() {}

    virtual solverPerformance solve
    (
        scalarField& psi,
        const scalarField& source,
        const direction cmpt = 0
    ) const
    {
        ldu_controls c;
        hipReadControls(controlDict_, SolverKind, c);

        // the log name the reference prints (PCG.C:73-77, GAMGSolverSolve.C:42, smoothSolver.C:85)
        word name(typeName);
        if (SolverKind == LDU_SOLVER_PCG || SolverKind == LDU_SOLVER_PBICG)
        {
            name = lduMatrix::preconditioner::getName(controlDict_) + typeName;
        }

        hipLduEntry& e = hipLookup(matrix_, interfaces_);
        hipSetCoeffs(e, matrix_, interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_);
        if
        (
            (SolverKind == LDU_SOLVER_GAMG || c.preconditioner == LDU_PRE_GAMG)
         && c.agglomerator == LDU_AGG_FACEAREAPAIR
        )
        {
            hipEnsureFaceWeights(e, matrix_);
        }

        hipDumpMatrix(fieldName_, matrix_, interfaces_, psi, source);
        ldu_perf perf;
        hipCheck(ldu_solve(e.mat, &c, psi.begin(), source.begin(), &perf, NULL), "hipLduSolver::solve");
        if (getenv("LDU_VERBOSE"))
        {
            Info<< "[hipLduSolvers] " << name << " for " << fieldName_ << " solved on the GPU in "
                << perf.solveSeconds << " s" << endl;
        }

        return solverPerformance
        (
            name, fieldName_, perf.initialResidual, perf.finalResidual, perf.nIterations,
            perf.converged, perf.singular
        );
    }
};

template<> const word hipLduSolver<LDU_SOLVER_PCG>::typeName("PCG");
template<> const word hipLduSolver<LDU_SOLVER_PBICG>::typeName("PBiCG");
template<> const word hipLduSolver<LDU_SOLVER_GAMG>::typeName("GAMG");
template<> const word hipLduSolver<LDU_SOLVER_SMOOTH>::typeName("smoothSolver");

typedef hipLduSolver<LDU_SOLVER_PCG> hipPCG;
typedef hipLduSolver<LDU_SOLVER_PBICG> hipPBiCG;
typedef hipLduSolver<LDU_SOLVER_GAMG> hipGAMG;
typedef hipLduSolver<LDU_SOLVER_SMOOTH> hipSmoothSolver;

// registration: stock names are overridden (set), hip* names added alongside
static const word nPCG("PCG"), nHipPCG("hipPCG"), nPBiCG("PBiCG"), nHipPBiCG("hipPBiCG"),
    nGAMG("GAMG"), nHipGAMG("hipGAMG"), nSmooth("smoothSolver"), nHipSmooth("hipSmoothSolver");

lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipPCG> addHipPCGSym_(nPCG);
lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipPCG> addHipPCGSym2_(nHipPCG);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipPBiCG> addHipPBiCGAsym_(nPBiCG);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipPBiCG> addHipPBiCGAsym2_(nHipPBiCG);
lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipGAMG> addHipGAMGSym_(nGAMG);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipGAMG> addHipGAMGAsym_(nGAMG);
lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipGAMG> addHipGAMGSym2_(nHipGAMG);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipGAMG> addHipGAMGAsym2_(nHipGAMG);
lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipSmoothSolver> addHipSmoothSym_(nSmooth);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipSmoothSolver> addHipSmoothAsym_(nSmooth);
lduMatrix::solver::addRemovablesymMatrixConstructorToTable<hipSmoothSolver> addHipSmoothSym2_(nHipSmooth);
lduMatrix::solver::addRemovableasymMatrixConstructorToTable<hipSmoothSolver> addHipSmoothAsym2_(nHipSmooth);


// ------------------------------------------------------------------ coupled solvers (`type coupled;`)
// LduMatrix<Type, scalar, scalar>::solver (fvMatrixSolve.C:222-277 builds the matrix): PCICG / PBiCCCG /
// PBiCICG / SmoothSolver resolve to ldu_coupled_solve; the stock entries are replaced, hip* names added.

static int hipCoupledPreconditionerKind(const word& n)
{
    if (n == "none") return LDU_CPRE_NONE;
    if (n == "diagonal") return LDU_CPRE_DIAGONAL;
    if (n == "DILU") return LDU_CPRE_DILU;
    FatalErrorIn("hipCoupledPreconditionerKind") << "coupled preconditioner " << n
        << " has no GPU implementation" << exit(FatalError);
    return -1;
}

template<class Type, int Kind>
class hipCoupledSolver
:
    public LduMatrix<Type, scalar, scalar>::solver
{
    typedef LduMatrix<Type, scalar, scalar> cMatrix;

public:

    static const word typeName;
    virtual const word& type() const { return typeName; }

    hipCoupledSolver(const word& fieldName, const cMatrix& matrix, const dictionary& solverDict)
    :
        cMatrix::solver(fieldName, matrix, solverDict)
    {}

    virtual ~hipCoupledSolver() {}

    virtual SolverPerformance<Type> solve(Field<Type>& psi) const
    {
        const cMatrix& M = this->matrix_;
        const direction nc = pTraits<Type>::nComponents;
        ldu_coupled_controls c;
        ldu_coupled_default_controls(&c, nc);
        c.solver = Kind;
        c.maxIter = this->maxIter_;
        for (direction k = 0; k < nc; k++)
        {
            c.tolerance[k] = component(this->tolerance_, k);
            c.relTol[k] = component(this->relTol_, k);
            // weight of component k in the Type's double inner product (PBiCCCG's gSumProd): unit_k && unit_k
            Type e(pTraits<Type>::zero);
            setComponent(e, k) = 1;
            c.innerProductWeights[k] = (e && e);
        }
        c.nSweeps = this->controlDict_.template lookupOrDefault<label>("nSweeps", 1);
        if (Kind == LDU_CSOLVER_SMOOTHSOLVER)
        {
            const word sm(this->controlDict_.lookup("smoother"));   // LduMatrixSmoother.C:41
            if (sm != "GaussSeidel")
            {
                FatalErrorIn("hipCoupledSolver::solve") << "coupled smoother " << sm
                    << " has no GPU implementation" << exit(FatalError);
            }
        }
        else
        {
            // LduMatrixPreconditioner.C:41: mandatory entry, read when the solver needs it
            c.preconditioner = hipCoupledPreconditionerKind(word(this->controlDict_.lookup("preconditioner")));
        }

        hipLduEntry& e = hipLookupAddr(M.lduAddr(), M.interfaces());
        hipCheck
        (
            ldu_matrix_set_coeffs
            (
                e.mat, M.diag().begin(), M.upper().begin(), M.asymmetric() ? M.lower().begin() : NULL
            ),
            "hipCoupledSolver::solve"
        );
        label k = 0;
        forAll(M.interfaces(), patchi)
        {
            if (M.interfaces().set(patchi))
            {
                hipCheck
                (
                    ldu_matrix_set_patch_coeffs
                    (
                        e.mat, k++, M.interfacesUpper()[patchi].begin(), M.interfacesLower()[patchi].begin()
                    ),
                    "hipCoupledSolver::solve"
                );
            }
        }

        ldu_coupled_perf perf;
        hipCheck
        (
            ldu_coupled_solve
            (
                e.mat, &c, reinterpret_cast<double*>(psi.begin()),
                reinterpret_cast<const double*>(M.source().begin()), &perf
            ),
            "hipCoupledSolver::solve"
        );
        if (getenv("LDU_VERBOSE"))
        {
            Info<< "[hipLduSolvers] coupled " << typeName << " for " << this->fieldName_
                << " solved on the GPU in " << perf.solveSeconds << " s, " << perf.nIterations
                << " iterations" << endl;
        }
        Type iR(pTraits<Type>::zero), fR(pTraits<Type>::zero);
        bool singular = true;
        for (direction k2 = 0; k2 < nc; k2++)
        {
            setComponent(iR, k2) = perf.initialResidual[k2];
            setComponent(fR, k2) = perf.finalResidual[k2];
            singular = singular && perf.singular[k2];
        }
        return SolverPerformance<Type>
        (
            typeName, this->fieldName_, iR, fR, perf.nIterations, perf.converged, singular
        );
    }
};

#define makeHipCoupledSolvers(Type)                                                                     \
    template<> const word hipCoupledSolver<Type, LDU_CSOLVER_PCICG>::typeName("PCICG");                 \
    template<> const word hipCoupledSolver<Type, LDU_CSOLVER_PBICCCG>::typeName("PBiCCCG");             \
    template<> const word hipCoupledSolver<Type, LDU_CSOLVER_PBICICG>::typeName("PBiCICG");             \
    template<> const word hipCoupledSolver<Type, LDU_CSOLVER_SMOOTHSOLVER>::typeName("SmoothSolver");   \
    LduMatrix<Type, scalar, scalar>::solver::addRemovablesymMatrixConstructorToTable                   \
        <hipCoupledSolver<Type, LDU_CSOLVER_PCICG> > addHipPCICG##Type##_(nPCICG);                      \
    LduMatrix<Type, scalar, scalar>::solver::addRemovablesymMatrixConstructorToTable                   \
        <hipCoupledSolver<Type, LDU_CSOLVER_PCICG> > addHipPCICG2##Type##_(nHipPCICG);                  \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_PBICCCG> > addHipPBiCCCG##Type##_(nPBiCCCG);                \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_PBICCCG> > addHipPBiCCCG2##Type##_(nHipPBiCCCG);            \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_PBICICG> > addHipPBiCICG##Type##_(nPBiCICG);                \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_PBICICG> > addHipPBiCICG2##Type##_(nHipPBiCICG);            \
    LduMatrix<Type, scalar, scalar>::solver::addRemovablesymMatrixConstructorToTable                   \
        <hipCoupledSolver<Type, LDU_CSOLVER_SMOOTHSOLVER> > addHipCSmoothSym##Type##_(nCSmooth);        \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_SMOOTHSOLVER> > addHipCSmoothAsym##Type##_(nCSmooth);       \
    LduMatrix<Type, scalar, scalar>::solver::addRemovablesymMatrixConstructorToTable                   \
        <hipCoupledSolver<Type, LDU_CSOLVER_SMOOTHSOLVER> > addHipCSmoothSym2##Type##_(nHipCSmooth);    \
    LduMatrix<Type, scalar, scalar>::solver::addRemovableasymMatrixConstructorToTable                  \
        <hipCoupledSolver<Type, LDU_CSOLVER_SMOOTHSOLVER> > addHipCSmoothAsym2##Type##_(nHipCSmooth);

static const word nPCICG("PCICG"), nHipPCICG("hipPCICG"), nPBiCCCG("PBiCCCG"), nHipPBiCCCG("hipPBiCCCG"),
    nPBiCICG("PBiCICG"), nHipPBiCICG("hipPBiCICG"), nCSmooth("SmoothSolver"), nHipCSmooth("hipSmoothSolver");

// the Types the reference instantiates (Solvers/lduSolvers.C:54-58)
makeHipCoupledSolvers(scalar)
makeHipCoupledSolvers(vector)
makeHipCoupledSolvers(sphericalTensor)
makeHipCoupledSolvers(symmTensor)
makeHipCoupledSolvers(tensor)


// ------------------------------------------------------------------ preconditioners (hipDIC, hipDILU)

template<int Kind>
class hipLduPreconditioner
:
    public lduMatrix::preconditioner
{
public:

    static const word typeName;
    virtual const word& type() const { return typeName; }

    hipLduPreconditioner(const lduMatrix::solver& sol, const dictionary&)
    :
        lduMatrix::preconditioner(sol)
    {
        hipLduEntry& e = hipLookup(sol.matrix(), sol.interfaces());
        hipSetCoeffs(e, sol.matrix(), sol.interfaceBouCoeffs(), sol.interfaceIntCoeffs(), sol.interfaces());
    }

    virtual void precondition(scalarField& wA, const scalarField& rA, const direction = 0) const
    {
        hipLduEntry& e = hipLookup(solver_.matrix(), solver_.interfaces());
        hipCheck(ldu_precondition(e.mat, Kind, wA.begin(), rA.begin(), 0), "hipLduPreconditioner");
    }

    virtual void preconditionT(scalarField& wT, const scalarField& rT, const direction = 0) const
    {
        hipLduEntry& e = hipLookup(solver_.matrix(), solver_.interfaces());
        hipCheck(ldu_precondition(e.mat, Kind, wT.begin(), rT.begin(), 1), "hipLduPreconditioner");
    }
};

template<> const word hipLduPreconditioner<LDU_PRE_DIC>::typeName("hipDIC");
template<> const word hipLduPreconditioner<LDU_PRE_DILU>::typeName("hipDILU");
typedef hipLduPreconditioner<LDU_PRE_DIC> hipDICPreconditioner;
typedef hipLduPreconditioner<LDU_PRE_DILU> hipDILUPreconditioner;

lduMatrix::preconditioner::addRemovablesymMatrixConstructorToTable<hipDICPreconditioner> addHipDIC_;
lduMatrix::preconditioner::addRemovableasymMatrixConstructorToTable<hipDILUPreconditioner> addHipDILU_;


// ------------------------------------------------------------------ smoother (hipGaussSeidel)

class hipGaussSeidelSmoother
:
    public lduMatrix::smoother
{
public:

    static const word typeName;
    virtual const word& type() const { return typeName; }

    hipGaussSeidelSmoother
    (
        const word& fieldName,
        const lduMatrix& matrix,
        const FieldField<Field, scalar>& interfaceBouCoeffs,
        const FieldField<Field, scalar>& interfaceIntCoeffs,
        const lduInterfaceFieldPtrsList& interfaces
    )
    :
        lduMatrix::smoother(fieldName, matrix, interfaceBouCoeffs, interfaceIntCoeffs, interfaces)
    {
        hipLduEntry& e = hipLookup(matrix_, interfaces_);
        hipSetCoeffs(e, matrix_, interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_);
    }

    virtual void smooth
    (
        scalarField& psi,
        const scalarField& source,
        const direction,
        const label nSweeps
    ) const
    {
        hipLduEntry& e = hipLookup(matrix_, interfaces_);
        hipCheck(ldu_smooth(e.mat, LDU_SM_GAUSSSEIDEL, psi.begin(), source.begin(), nSweeps),
                 "hipGaussSeidelSmoother");
    }
};

const word hipGaussSeidelSmoother::typeName("hipGaussSeidel");
lduMatrix::smoother::addRemovablesymMatrixConstructorToTable<hipGaussSeidelSmoother> addHipGSSym_;
lduMatrix::smoother::addRemovableasymMatrixConstructorToTable<hipGaussSeidelSmoother> addHipGSAsym_;

} // End namespace Foam
