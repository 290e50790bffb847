// Mesh side of the path: what turns the polyMesh arrays (points, faces, owner,
// neighbour) into the geometric fields the fv stencils and the GAMG agglomeration read - face centres and area
// vectors, cell centres and volumes (primitiveMeshFaceCentresAndAreas.C:73-131,
// primitiveMeshCellCentresAndVols.C:72-147), linear interpolation weights and deltaCoeffs
// (surfaceInterpolation.C:163-185, :227-231) - and the reference's cell renumbering
// (bandCompression.C:43-146, what renumberMesh applies).  Geometry: one thread per face / per cell, every
// sum in the reference's order, no FMA contraction: bit-identical to libOpenFOAM.  Renumbering: host code.
#include <algorithm>
#include <deque>
#include <vector>

#include "ldu_internal.hpp"

#define BLK 256
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int ewGrid(long n) { int g = cdiv(n, BLK); return g < 1 ? 1 : (g > 4096 ? 4096 : g); }

static const double kVSmall = 1e-300;       // VSMALL (doubleScalar.H)
static const double kRootVSmall = 1e-150;   // ROOTVSMALL

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 ld3(const double* p, long i) { return V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ void st3(double* p, long i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 mul(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ V3 divs(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   // VectorI.H:152-155
__device__ __forceinline__ V3 cross(V3 a, V3 b)                                                      // VectorI.H:159-168
{
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double mag(V3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// primitiveMeshFaceCentresAndAreas.C:73-131
__global__ void __launch_bounds__(BLK)
mesh_face_kernel(int nFaces, const double* __restrict__ p, const int* __restrict__ fStart,
                 const int* __restrict__ fPts, double* __restrict__ fCtrs, double* __restrict__ fAreas)
{
    for (int facei = blockIdx.x * BLK + threadIdx.x; facei < nFaces; facei += gridDim.x * BLK)
    {
        const int* f = fPts + fStart[facei];
        const int nPoints = fStart[facei + 1] - fStart[facei];
        if (nPoints == 3)
        {
            const V3 p0 = ld3(p, f[0]), p1 = ld3(p, f[1]), p2 = ld3(p, f[2]);
            st3(fCtrs, facei, mul(1.0 / 3.0, add(add(p0, p1), p2)));
            st3(fAreas, facei, mul(0.5, cross(sub(p1, p0), sub(p2, p0))));
            continue;
        }
        V3 sumN{0, 0, 0}, sumAc{0, 0, 0};
        double sumA = 0.0;
        V3 fCentre = ld3(p, f[0]);
        for (int pi = 1; pi < nPoints; pi++) fCentre = add(fCentre, ld3(p, f[pi]));
        fCentre = divs(fCentre, (double)nPoints);
        for (int pi = 0; pi < nPoints; pi++)
        {
            const V3 cur = ld3(p, f[pi]);
            const V3 nextPoint = ld3(p, f[(pi + 1) % nPoints]);
            const V3 c = add(add(cur, nextPoint), fCentre);
            const V3 n = cross(sub(nextPoint, cur), sub(fCentre, cur));
            const double a = mag(n);
            sumN = add(sumN, n);
            sumA += a;
            sumAc = add(sumAc, mul(a, c));
        }
        if (sumA < kRootVSmall)
        {
            st3(fCtrs, facei, fCentre);
            st3(fAreas, facei, V3{0, 0, 0});
        }
        else
        {
            st3(fCtrs, facei, divs(mul(1.0 / 3.0, sumAc), sumA));
            st3(fAreas, facei, mul(0.5, sumN));
        }
    }
}

// primitiveMeshCellCentresAndVols.C:72-147 as a gather per cell: its owned faces in face order, then the faces
// it is neighbour of in face order - the order in which the two face loops of the reference reach the cell
__global__ void __launch_bounds__(BLK)
mesh_cell_kernel(int nCells, const int* __restrict__ cStart, const int* __restrict__ cFaces, const int* __restrict__ nOwn,
                 const double* __restrict__ fCtrs, const double* __restrict__ fAreas, double* __restrict__ cellCtrs,
                 double* __restrict__ cellVols)
{
    for (int celli = blockIdx.x * BLK + threadIdx.x; celli < nCells; celli += gridDim.x * BLK)
    {
        const int b = cStart[celli], e = cStart[celli + 1], no = nOwn[celli];
        V3 cEst{0, 0, 0};
        for (int i = b; i < e; i++) cEst = add(cEst, ld3(fCtrs, cFaces[i]));
        cEst = divs(cEst, (double)(e - b));
        V3 ctr{0, 0, 0};
        double vol = 0.0;
        for (int i = b; i < e; i++)
        {
            const int facei = cFaces[i];
            const V3 fc = ld3(fCtrs, facei), fa = ld3(fAreas, facei);
            const double d = (i - b < no) ? dot(fa, sub(fc, cEst)) : dot(fa, sub(cEst, fc));
            const double pyr3Vol = d > kVSmall ? d : kVSmall;   // max(d, VSMALL)
            const V3 pc = add(mul(3.0 / 4.0, fc), mul(1.0 / 4.0, cEst));
            ctr = add(ctr, mul(pyr3Vol, pc));
            vol += pyr3Vol;
        }
        st3(cellCtrs, celli, divs(ctr, vol));
        cellVols[celli] = vol * (1.0 / 3.0);
    }
}

// surfaceInterpolation.C:163-185 (weights), :227-231 (deltaCoeffs); fvMesh::magSf = mag(Sf) + VSMALL (fvMeshGeometry.C:113)
__global__ void __launch_bounds__(BLK)
mesh_factors_kernel(int nInternal, const int* __restrict__ owner, const int* __restrict__ neighbour,
                    const double* __restrict__ Cf, const double* __restrict__ Sf, const double* __restrict__ C,
                    double* __restrict__ w, double* __restrict__ delta, double* __restrict__ magSf)
{
    for (int facei = blockIdx.x * BLK + threadIdx.x; facei < nInternal; facei += gridDim.x * BLK)
    {
        const V3 sf = ld3(Sf, facei), cf = ld3(Cf, facei);
        const V3 co = ld3(C, owner[facei]), cn = ld3(C, neighbour[facei]);
        const double SfdOwn = fabs(dot(sf, sub(cf, co)));
        const double SfdNei = fabs(dot(sf, sub(cn, cf)));
        if (w) w[facei] = SfdNei / (SfdOwn + SfdNei);
        if (delta) delta[facei] = 1.0 / mag(sub(cn, co));
        if (magSf) magSf[facei] = mag(sf) + kVSmall;
    }
}

// faceAreaPairGAMGAgglomeration.C:48-73: mag(cmptMultiply(Sf/sqrt(magSf), vector(1, 1.01, 1.02))) with
// magSf = mag(Sf) + VSMALL (fvMeshGeometry.C:101-114)
__global__ void __launch_bounds__(BLK)
mesh_faceAreaPair_kernel(int nInternal, const double* __restrict__ Sf, double* __restrict__ w)
{
    for (int facei = blockIdx.x * BLK + threadIdx.x; facei < nInternal; facei += gridDim.x * BLK)
    {
        const V3 sf = ld3(Sf, facei);
        const double r = sqrt(mag(sf) + kVSmall);
        V3 t;
        t.x = sf.x / r; t.y = (sf.y / r) * 1.01; t.z = (sf.z / r) * 1.02;
        w[facei] = mag(t);
    }
}

template <class T>
struct DevBuf {
    T* d = nullptr;
    bool own = false;
    T* host = nullptr;   // copy back here on release
    size_t n = 0;
    ~DevBuf() { if (own && d) (void)hipFree(d); }
};

static bool on_device(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

template <class T>
static int dev_in(DevBuf<T>& b, const T* user, size_t n, hipStream_t s)
{
    b.n = n;
    if (on_device(user)) { b.d = const_cast<T*>(user); return 0; }
    LDU_CHECK_HIP(hipMalloc((void**)&b.d, sizeof(T) * (n ? n : 1)));
    b.own = true;
    if (n) LDU_CHECK_HIP(hipMemcpyAsync(b.d, user, sizeof(T) * n, hipMemcpyHostToDevice, s));
    return 0;
}
template <class T>
static int dev_out(DevBuf<T>& b, T* user, size_t n)
{
    b.n = n;
    if (!user) return 0;
    if (on_device(user)) { b.d = user; return 0; }
    LDU_CHECK_HIP(hipMalloc((void**)&b.d, sizeof(T) * (n ? n : 1)));
    b.own = true;
    b.host = user;
    return 0;
}
template <class T>
static int dev_back(DevBuf<T>& b, hipStream_t s)
{
    if (b.host && b.n) LDU_CHECK_HIP(hipMemcpyAsync(b.host, b.d, sizeof(T) * b.n, hipMemcpyDeviceToHost, s));
    return 0;
}

extern "C" {

int ldu_mesh_geometry(ldu_ctx* ctx, int32_t nPoints, const double* points, int32_t nFaces, const int32_t* faceStart,
                      const int32_t* facePoints, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                      const int32_t* neighbour, double* faceCentres, double* faceAreas, double* cellCentres,
                      double* cellVolumes)
{
    if (!ctx || nPoints < 0 || nFaces < 0 || nCells < 0 || nInternalFaces < 0 || nInternalFaces > nFaces)
    {
        ldu_set_error("ldu_mesh_geometry: bad sizes");
        return -14;
    }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // host copies of the addressing for the validity checks and the cell -> faces table
    std::vector<int> hStart(nFaces + 1), hOwn(nFaces), hNei(nInternalFaces);
    LDU_CHECK_HIP(hipMemcpy(hStart.data(), faceStart, sizeof(int) * (nFaces + 1), hipMemcpyDefault));
    if (nFaces) LDU_CHECK_HIP(hipMemcpy(hOwn.data(), owner, sizeof(int) * nFaces, hipMemcpyDefault));
    if (nInternalFaces) LDU_CHECK_HIP(hipMemcpy(hNei.data(), neighbour, sizeof(int) * nInternalFaces, hipMemcpyDefault));
    const int nFP = hStart[nFaces];
    for (int f = 0; f < nFaces; f++)
    {
        if (hStart[f + 1] - hStart[f] < 3) { ldu_set_error("ldu_mesh_geometry: face with fewer than 3 points"); return -14; }
        if (hOwn[f] < 0 || hOwn[f] >= nCells || (f < nInternalFaces && (hNei[f] < 0 || hNei[f] >= nCells)))
        {
            ldu_set_error("ldu_mesh_geometry: owner / neighbour out of range");
            return -14;
        }
    }
    std::vector<int> hPts(nFP);
    if (nFP) LDU_CHECK_HIP(hipMemcpy(hPts.data(), facePoints, sizeof(int) * nFP, hipMemcpyDefault));
    for (int i = 0; i < nFP; i++)
        if (hPts[i] < 0 || hPts[i] >= nPoints) { ldu_set_error("ldu_mesh_geometry: point label out of range"); return -14; }
    // cell -> faces: owned faces ascending, then neighbour faces ascending
    std::vector<int> cStart(nCells + 1, 0), nOwn(nCells, 0);
    for (int f = 0; f < nFaces; f++) { cStart[hOwn[f] + 1]++; nOwn[hOwn[f]]++; }
    for (int f = 0; f < nInternalFaces; f++) cStart[hNei[f] + 1]++;
    for (int c = 0; c < nCells; c++) cStart[c + 1] += cStart[c];
    std::vector<int> cFaces(cStart[nCells]), pos(cStart.begin(), cStart.end() - 1);
    for (int f = 0; f < nFaces; f++) cFaces[pos[hOwn[f]]++] = f;
    for (int f = 0; f < nInternalFaces; f++) cFaces[pos[hNei[f]]++] = f;
    for (int c = 0; c < nCells; c++)
        if (cStart[c + 1] == cStart[c]) { ldu_set_error("ldu_mesh_geometry: cell without faces"); return -14; }

    DevBuf<double> dP, dCf, dSf, dC, dV;
    DevBuf<int> dStart, dPts, dCStart, dCFaces, dNOwn;
    if (dev_in(dP, points, (size_t)3 * nPoints, s) || dev_in(dStart, hStart.data(), (size_t)nFaces + 1, s)
        || dev_in(dPts, hPts.data(), (size_t)nFP, s) || dev_in(dCStart, cStart.data(), (size_t)nCells + 1, s)
        || dev_in(dCFaces, cFaces.data(), cFaces.size(), s) || dev_in(dNOwn, nOwn.data(), (size_t)nCells, s))
        return -1;
    if (!faceCentres || !faceAreas || !cellCentres || !cellVolumes)
    {
        ldu_set_error("ldu_mesh_geometry: all four outputs are required");
        return -14;
    }
    if (dev_out(dCf, faceCentres, (size_t)3 * nFaces) || dev_out(dSf, faceAreas, (size_t)3 * nFaces)
        || dev_out(dC, cellCentres, (size_t)3 * nCells) || dev_out(dV, cellVolumes, (size_t)nCells))
        return -1;
    if (nFaces)
        mesh_face_kernel<<<ewGrid(nFaces), BLK, 0, s>>>(nFaces, dP.d, dStart.d, dPts.d, dCf.d, dSf.d);
    if (nCells)
        mesh_cell_kernel<<<ewGrid(nCells), BLK, 0, s>>>(nCells, dCStart.d, dCFaces.d, dNOwn.d, dCf.d, dSf.d, dC.d, dV.d);
    LDU_CHECK_HIP(hipGetLastError());
    if (dev_back(dCf, s) || dev_back(dSf, s) || dev_back(dC, s) || dev_back(dV, s)) return -1;
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int ldu_mesh_interpolation_factors(ldu_ctx* ctx, int32_t nCells, int32_t nInternalFaces, const int32_t* owner,
                                   const int32_t* neighbour, const double* faceCentres, const double* faceAreas,
                                   const double* cellCentres, double* weights, double* deltaCoeffs, double* magSf)
{
    if (!ctx || nInternalFaces < 0) { ldu_set_error("ldu_mesh_interpolation_factors: bad sizes"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    DevBuf<int> dO, dN;
    DevBuf<double> dCf, dSf, dC, dW, dD, dM;
    const size_t nF = (size_t)nInternalFaces;
    if (dev_in(dO, owner, nF, s) || dev_in(dN, neighbour, nF, s) || dev_in(dCf, faceCentres, 3 * nF, s)
        || dev_in(dSf, faceAreas, 3 * nF, s) || dev_in(dC, cellCentres, 3 * (size_t)nCells, s)
        || dev_out(dW, weights, nF) || dev_out(dD, deltaCoeffs, nF) || dev_out(dM, magSf, nF))
        return -1;
    if (nF)
        mesh_factors_kernel<<<ewGrid(nInternalFaces), BLK, 0, s>>>(nInternalFaces, dO.d, dN.d, dCf.d, dSf.d, dC.d, dW.d,
                                                                  dD.d, dM.d);
    LDU_CHECK_HIP(hipGetLastError());
    if (dev_back(dW, s) || dev_back(dD, s) || dev_back(dM, s)) return -1;
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int ldu_addr_set_face_areas(ldu_addr* a, const double* Sf)
{
    if (!a || (!Sf && a->nFaces)) { ldu_set_error("ldu_addr_set_face_areas: null argument"); return -14; }
    LDU_CHECK_HIP(hipSetDevice(a->ctx->device));
    hipStream_t s = a->ctx->stream;
    const size_t nF = (size_t)a->nFaces;
    a->faceWeights.resize(nF);
    if (!nF) return 0;
    DevBuf<double> dSf, dW;
    if (dev_in(dSf, Sf, 3 * nF, s) || dev_out(dW, a->faceWeights.data(), nF)) return -1;
    mesh_faceAreaPair_kernel<<<ewGrid(a->nFaces), BLK, 0, s>>>(a->nFaces, dSf.d, dW.d);
    LDU_CHECK_HIP(hipGetLastError());
    if (dev_back(dW, s)) return -1;
    LDU_CHECK_HIP(hipStreamSynchronize(s));
    return 0;
}

int ldu_addr_get_face_weights(const ldu_addr* a, double* w)
{
    if ((int)a->faceWeights.size() != a->nFaces) { ldu_set_error("no face weights set"); return -7; }
    if (a->nFaces) LDU_CHECK_HIP(hipMemcpy(w, a->faceWeights.data(), sizeof(double) * a->nFaces, hipMemcpyDefault));
    return 0;
}

int ldu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// bandCompression.C:43-146 on the cell-cell addressing of the internal faces (primitiveMeshCellCells.C: face
// order, the owner learns the neighbour and the neighbour the owner).  Literal behaviours kept: every restart
// picks the unvisited cell with the FEWEST neighbours (first one on ties); the unvisited neighbours of a cell
// are queued in their cellCells order - the reference sorts an index list by connectivity (:131) but then
// appends nbrs[i], not nbrs[order[i]] (:134-137).  Host code: it runs once per mesh.
// nParts compact sub-domains of (nearly) equal size for any numbering (sub-domain mode, ldu_addr_set_subdomains): breadth-first
// blobs of ceil(nCells / nParts) cells grown from the lowest unassigned cell - under a bandwidth-reducing numbering they tile
// the shells of the numbering -, pockets a blob leaves behind join the neighbouring part with the fewest cells.  Host code.
int ldu_partition_blobs(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr, int32_t nParts,
                        int32_t* part)
{
    return partition_blobs(nCells, nFaces, lowerAddr, upperAddr, nParts, part) < 0 ? -2 : 0;
}

int ldu_partition_blobs_footprint(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                                  int64_t slotTarget, int32_t maxParts, int32_t* part, int32_t* nParts)
{
    const int n = partition_blobs_slots(nCells, nFaces, lowerAddr, upperAddr, (long)slotTarget, maxParts, part);
    if (n < 0) return -2;
    if (nParts) *nParts = n;
    return 0;
}

// (also the block engine's partitioner, ldu_blocks.hip) -> parts actually made (<= nParts), -1 on bad arguments
int partition_blobs(int nCells, int nFaces, const int* lowerAddr, const int* upperAddr, int nParts, int* part)
{
    if (nParts < 1 || nCells < 0) { ldu_set_error("ldu_partition_blobs: nParts >= 1"); return -1; }
    std::vector<int> start((size_t)nCells + 1, 0), adj(2 * (size_t)nFaces);
    for (int f = 0; f < nFaces; f++) { start[lowerAddr[f] + 1]++; start[upperAddr[f] + 1]++; }
    for (int c = 0; c < nCells; c++) start[c + 1] += start[c];
    {
        std::vector<int> pos(start.begin(), start.end() - 1);
        for (int f = 0; f < nFaces; f++) { adj[pos[lowerAddr[f]]++] = upperAddr[f]; adj[pos[upperAddr[f]]++] = lowerAddr[f]; }
    }
    for (int c = 0; c < nCells; c++) part[c] = -1;
    const long target = ((long)nCells + nParts - 1) / nParts;
    std::vector<long> size(nParts, 0);
    std::vector<int> q;
    std::vector<char> pocket((size_t)nCells, 0);   // cells of a blob that got stuck small: left for the pocket pass below
    int seed = 0;
    for (int r = 0; r < nParts; r++)
    {
        for (;;)
        {
            while (seed < nCells && (part[seed] >= 0 || pocket[seed])) seed++;
            if (seed >= nCells) break;
            const long lim = r == nParts - 1 ? nCells : target;
            q.clear();
            q.push_back(seed);
            part[seed] = r;
            size_t h = 0;
            while (h < q.size() && (long)q.size() < lim)
            {
                const int c = q[h++];
                for (int t = start[c]; t < start[c + 1] && (long)q.size() < lim; t++)
                    if (part[adj[t]] < 0 && !pocket[adj[t]]) { part[adj[t]] = r; q.push_back(adj[t]); }
            }
            if ((long)q.size() >= target / 2 || (long)q.size() == lim) { size[r] = (long)q.size(); break; }
            // enclosed by earlier blobs before it reached half its size: not a sub-domain of its own
            for (int c : q) { part[c] = -1; pocket[c] = 1; }
        }
    }
    // pockets: repeatedly the unassigned cells that touch a part join the smallest part they touch
    for (;;)
    {
        long left = 0, moved = 0;
        for (int c = 0; c < nCells; c++)
        {
            if (part[c] >= 0) continue;
            left++;
            int best = -1;
            for (int t = start[c]; t < start[c + 1]; t++)
            {
                const int b = part[adj[t]];
                if (b >= 0 && (best < 0 || size[b] < size[best])) best = b;
            }
            if (best >= 0) { part[c] = best; size[best]++; moved++; }
        }
        if (!left) break;
        if (!moved)
        {
            // a component no part touches: the smallest part takes it
            int sm = 0;
            for (int r = 1; r < nParts; r++) if (size[r] < size[sm]) sm = r;
            for (int c = 0; c < nCells; c++) if (part[c] < 0) { part[c] = sm; size[sm]++; }
            break;
        }
    }
    // (the last seeds may have found nothing but pockets: labels are compacted, fewer than nParts parts can come back)
    std::vector<int> lab(nParts, -1);
    int used = 0;
    for (int r = 0; r < nParts; r++) if (size[r] > 0) lab[r] = used++;
    if (used < nParts) for (int c = 0; c < nCells; c++) part[c] = lab[part[c]];
    return used;
}

// The block engine's partitioner for levels whose equal-SIZE blobs do not fit (ldu_blocks.hip): breadth-first blobs of equal
// FOOTPRINT - a blob grows until its cells plus the distinct cells outside it that touch it (its ghosts, whichever part they end
// up in) reach slotTarget.  Equal cell counts leave the footprints 1.3-1.4 x apart (the surface of a blob in an agglomerated level
// varies a lot), and the footprint of the LARGEST block is what has to fit into a workgroup's LDS.  As many parts as it takes (at
// most maxParts, else -2); pockets join the neighbouring part of the smallest footprint.  -> parts made
int partition_blobs_slots(int nCells, int nFaces, const int* lowerAddr, const int* upperAddr, long slotTarget, int maxParts, int* part)
{
    if (slotTarget < 8 || maxParts < 1 || nCells < 0) { ldu_set_error("partition_blobs_slots: bad arguments"); return -1; }
    std::vector<int> start((size_t)nCells + 1, 0), adj(2 * (size_t)nFaces);
    for (int f = 0; f < nFaces; f++) { start[lowerAddr[f] + 1]++; start[upperAddr[f] + 1]++; }
    for (int c = 0; c < nCells; c++) start[c + 1] += start[c];
    {
        std::vector<int> pos(start.begin(), start.end() - 1);
        for (int f = 0; f < nFaces; f++) { adj[pos[lowerAddr[f]]++] = upperAddr[f]; adj[pos[upperAddr[f]]++] = lowerAddr[f]; }
    }
    for (int c = 0; c < nCells; c++) part[c] = -1;
    std::vector<long> foot;                          // cells + ghosts of a part
    std::vector<int> q;
    std::vector<int> seen((size_t)nCells, -1);       // seen[n] == r: n has been counted as a ghost of blob r
    std::vector<char> pocket((size_t)nCells, 0);
    int seed = 0, r = 0;
    for (;; )
    {
        while (seed < nCells && (part[seed] >= 0 || pocket[seed])) seed++;
        if (seed >= nCells) break;
        if (r >= maxParts) { ldu_set_error("partition_blobs_slots: more parts than allowed"); return -2; }
        q.clear();
        long ghosts = 0;
        auto take = [&](int c) {
            part[c] = r;
            q.push_back(c);
            if (seen[c] == r) ghosts--;              // was counted as a ghost: a cell of the blob now
            for (int t = start[c]; t < start[c + 1]; t++)
            {
                const int n = adj[t];
                if (part[n] != r && seen[n] != r) { seen[n] = r; ghosts++; }
            }
        };
        take(seed);
        size_t h = 0;
        bool full = false;
        while (h < q.size() && !full)
        {
            const int c = q[h++];
            for (int t = start[c]; t < start[c + 1]; t++)
            {
                const int n = adj[t];
                if (part[n] >= 0 || pocket[n]) continue;
                if ((long)q.size() + ghosts + (long)(start[n + 1] - start[n]) > slotTarget) { full = true; break; }
                take(n);
            }
        }
        if (full || (long)q.size() + ghosts >= slotTarget / 6)
        {
            foot.push_back((long)q.size() + ghosts);
            r++;
            continue;
        }
        // enclosed by earlier blobs before it reached a sixth of a footprint: not a block of its own (a larger remainder is: the
        // parts around it would have to take it in whatever their own footprint)
        for (int c : q) { part[c] = -1; pocket[c] = 1; }
        for (int c : q) seen[c] = -1;
    }
    const int nParts = r;
    if (nParts == 0)
    {
        // nothing but pockets (a tiny graph): one part
        for (int c = 0; c < nCells; c++) part[c] = 0;
        return nCells ? 1 : 0;
    }
    for (;;)
    {
        long left = 0, moved = 0;
        for (int c = 0; c < nCells; c++)
        {
            if (part[c] >= 0) continue;
            left++;
            int best = -1;
            for (int t = start[c]; t < start[c + 1]; t++)
            {
                const int b = part[adj[t]];
                if (b >= 0 && (best < 0 || foot[b] < foot[best])) best = b;
            }
            if (best >= 0) { part[c] = best; foot[best] += 1 + (start[c + 1] - start[c]) / 2; moved++; }
        }
        if (!left) break;
        if (!moved)
        {
            int sm = 0;
            for (int p = 1; p < nParts; p++) if (foot[p] < foot[sm]) sm = p;
            for (int c = 0; c < nCells; c++) if (part[c] < 0) { part[c] = sm; foot[sm]++; }
            break;
        }
    }
    return nParts;
}

int ldu_band_compression(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                         int32_t* newOrder)
{
    if (nCells < 0 || nFaces < 0) { ldu_set_error("ldu_band_compression: bad sizes"); return -14; }
    std::vector<int> start(nCells + 1, 0);
    for (int f = 0; f < nFaces; f++)
    {
        const int l = lowerAddr[f], u = upperAddr[f];
        if (l < 0 || u < 0 || l >= nCells || u >= nCells) { ldu_set_error("ldu_band_compression: address out of range"); return -14; }
        start[l + 1]++;
        start[u + 1]++;
    }
    for (int c = 0; c < nCells; c++) start[c + 1] += start[c];
    std::vector<int> cc(start[nCells]), pos(start.begin(), start.end() - 1);
    for (int f = 0; f < nFaces; f++)
    {
        cc[pos[lowerAddr[f]]++] = upperAddr[f];
        cc[pos[upperAddr[f]]++] = lowerAddr[f];
    }
    std::vector<char> visited(nCells, 0);
    // restart candidates: cells ordered by (number of neighbours, label); a cursor skips the visited ones
    std::vector<int> byWeight(nCells);
    for (int c = 0; c < nCells; c++) byWeight[c] = c;
    std::stable_sort(byWeight.begin(), byWeight.end(),
                     [&](int a, int b) { return start[a + 1] - start[a] < start[b + 1] - start[b]; });
    size_t cursor = 0;
    int cellInOrder = 0;
    std::deque<int> nextCell;
    for (;;)
    {
        while (cursor < byWeight.size() && visited[byWeight[cursor]]) cursor++;
        if (cursor == byWeight.size()) break;
        nextCell.push_back(byWeight[cursor]);
        while (!nextCell.empty())
        {
            const int cur = nextCell.front();
            nextCell.pop_front();
            if (visited[cur]) continue;
            visited[cur] = 1;
            newOrder[cellInOrder++] = cur;
            for (int i = start[cur]; i < start[cur + 1]; i++)
                if (!visited[cc[i]]) nextCell.push_back(cc[i]);
        }
    }
    return 0;
}

// What renumberMesh does to the matrix addressing with such an order (cell newOrder[i] becomes cell i): faces
// are re-oriented so that lower < upper and sorted into upper-triangular order (owner, then neighbour:
// lduAddressing.C:92-126 relies on it).  faceMap[newFace] = old face, flip[newFace] = 1 when the face changed
// orientation (its lower / upper coefficients swap, a flux changes sign).
int ldu_renumber_addressing(int32_t nCells, int32_t nFaces, const int32_t* lowerAddr, const int32_t* upperAddr,
                            const int32_t* newOrder, int32_t* newLower, int32_t* newUpper, int32_t* faceMap,
                            uint8_t* flip)
{
    std::vector<int> rev(nCells, -1);
    for (int i = 0; i < nCells; i++)
    {
        if (newOrder[i] < 0 || newOrder[i] >= nCells || rev[newOrder[i]] != -1)
        {
            ldu_set_error("ldu_renumber_addressing: newOrder is not a permutation");
            return -14;
        }
        rev[newOrder[i]] = i;
    }
    struct F { int l, u, old; unsigned char flip; };
    std::vector<F> fs(nFaces);
    for (int f = 0; f < nFaces; f++)
    {
        const int a = rev[lowerAddr[f]], b = rev[upperAddr[f]];
        fs[f] = a < b ? F{a, b, f, 0} : F{b, a, f, 1};
    }
    std::stable_sort(fs.begin(), fs.end(), [](const F& x, const F& y) { return x.l != y.l ? x.l < y.l : x.u < y.u; });
    for (int f = 0; f < nFaces; f++)
    {
        newLower[f] = fs[f].l;
        newUpper[f] = fs[f].u;
        if (faceMap) faceMap[f] = fs[f].old;
        if (flip) flip[f] = fs[f].flip;
    }
    return 0;
}

// A cell numbering for the sweep engines (DESIGN "Numbering"): `order` (normally Foam::bandCompression's) cut into tiles of
// tileSize consecutive cells, the tiles put into the order of a hash of (seed, tile index), the order inside a tile kept.
// A GaussSeidel sweep in this numbering still runs through every tile in bandCompression's order - a tile is a patch of one
// or two breadth-first shells - but the longest chain of cells that must be smoothed one after the other no longer spans the
// mesh: it spans a handful of tiles (the longest increasing path through randomly ranked tiles), and the pair matching of the
// agglomeration (pairGAMGAgglomerate.C:83-197) hands the same property down to every coarse level, whose numbering is the
// order in which it reaches the fine cells.  Deterministic (splitmix64), the same on every host.
int ldu_tile_shuffle(int32_t nCells, const int32_t* order, int32_t tileSize, uint64_t seed, int32_t* newOrder)
{
    if (nCells < 0 || tileSize < 1) { ldu_set_error("ldu_tile_shuffle: bad sizes"); return -14; }
    const long nT = ((long)nCells + tileSize - 1) / tileSize;
    std::vector<std::pair<uint64_t, long>> key((size_t)nT);
    for (long t = 0; t < nT; t++)
    {
        uint64_t z = (uint64_t)t + (seed << 40) + 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        key[(size_t)t] = {z ^ (z >> 31), t};
    }
    std::sort(key.begin(), key.end());
    long o = 0;
    for (long i = 0; i < nT; i++)
    {
        const long t = key[(size_t)i].second;
        const long a = t * tileSize, b = std::min<long>(a + tileSize, nCells);
        for (long c = a; c < b; c++) newOrder[o++] = order ? order[c] : (int32_t)c;
    }
    return 0;
}

}  // extern "C"
