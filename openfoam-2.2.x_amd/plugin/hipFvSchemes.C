/*---------------------------------------------------------------------------*\
  libhipFvSchemes.so - the finite-volume stencils of libldugpu behind OpenFOAM-2.2.x's own run-time selection tables of
  libfiniteVolume.

  Loaded with   libs ("libhipLduSolvers.so" "libhipFvSchemes.so");   in system/controlDict it adds the scheme name
  `hipGauss` next to the stock `Gauss` in three tables; a case selects it in system/fvSchemes, e.g.

      gradSchemes      { default hipGauss linear; }
      divSchemes       { div(phi,U) bounded hipGauss upwind; }
      laplacianSchemes { default hipGauss linear corrected; }

  No source of the reference changes; an unchanged simpleFoam / icoFoam then runs

      fv::laplacianScheme<Type, scalar>   (laplacianScheme.H:97-104, makeFvLaplacianTypeScheme :214-233)
          hipGaussLaplacianScheme::fvmLaplacian   -> ldu_fvm_laplacian       (gaussLaplacianScheme.C:46-88: upper, negSumDiag)
      fv::convectionScheme<Type>          (convectionScheme.H:82-95, makeFvConvectionTypeScheme :206-211)
          hipGaussConvectionScheme::fvmDiv        -> ldu_fvm_div             (gaussConvectionScheme.C:68-107: lower, upper, negSumDiag)
      fv::gradScheme<Type>                (gradScheme.H:87, makeFvGradTypeScheme :191-196)
          hipGaussGrad::calcGrad                  -> ldu_fvc_gaussGradFull   (gaussGrad.C:41-110: face loop, patch faces, / V)

  on the device.  Everything else of a scheme (the boundary coefficients of the patch fields, the explicit non-orthogonal
  correction, the interpolation weights of the chosen interpolation scheme, correctBoundaryConditions) is the base class's
  own code: the classes derive from the reference's gaussLaplacianScheme / gaussConvectionScheme / gaussGrad and replace
  the face loops only.  The kernels reproduce those loops bit for bit (tests/test_gpu_fv.py against vectors of the
  reference's libfiniteVolume), so a run with `hipGauss` prints the solver log of the run with `Gauss`
  (tests/test_simplefoam_pitzdaily.py::test_simplefoam_with_hipgauss_schemes).

  What this does NOT do: keep the assembled coefficients on the device for the solve.  OpenFOAM's Field has no access
  hook, a tmp<fvMatrix> is freely modified between assembly and solve (operator+=, relax, setReference), and hashing the
  host arrays to validate a device mirror costs more than uploading them: the matrix a scheme
  returns lives in host memory like the stock scheme's, and the solver plug-in uploads what it is handed.
\*---------------------------------------------------------------------------*/

#include "fvMesh.H"
#include "fvMatrices.H"
#include "volFields.H"
#include "surfaceFields.H"
#include "gaussLaplacianScheme.H"
#include "gaussConvectionScheme.H"
#include "gaussGrad.H"
#include "zeroGradientFvPatchField.H"
#include "fvcDiv.H"
#include "fvcSurfaceIntegrate.H"

#include "ldugpu.h"

#include <map>
#include <vector>
#include <stdint.h>

namespace Foam
{

ldu_ctx* hipLduSharedContext();     // hipLduSolvers.C: the process's one device context (created on first use)

// ------------------------------------------------------------------ per-mesh device addressing

// lduAddressing of the mesh (owner / neighbour of the internal faces) and its patches (sizes, faceCells, coupled) on the
// device, built on first use and rebuilt when the mesh says its topology changed or the lists moved.
struct hipFvMeshEntry
{
    ldu_addr* addr;
    ldu_fv_boundary* bnd;
    label nCells, nFaces, nPatchFaces;
    const label* lPtr;
    const label* uPtr;
    uint64_t hash;          // owner / neighbour / faceCells: a strided sample, every entry while the mesh says it is changing
};

static uint64_t hipFvHash(const fvMesh& mesh, const bool full)
{
    uint64_t h = 0x243F6A8885A308D3ULL;
    const labelUList* lists[2] = {&mesh.lduAddr().lowerAddr(), &mesh.lduAddr().upperAddr()};
    for (int t = 0; t < 2; t++)
    {
        const labelUList& l = *lists[t];
        const label n = l.size(), step = (!full && n > 4096) ? n/4096 : 1;
        for (label i = 0; i < n; i += step) h ^= uint64_t(l[i]) + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    }
    forAll(mesh.boundary(), patchi)
    {
        const labelUList& fc = mesh.boundary()[patchi].faceCells();
        const label n = fc.size(), step = (!full && n > 1024) ? n/1024 : 1;
        h ^= uint64_t(n) + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
        for (label i = 0; i < n; i += step) h ^= uint64_t(fc[i]) + 0x9E3779B97F4A7C15ULL + (h << 6) + (h >> 2);
    }
    return h;
}

static std::map<const fvMesh*, hipFvMeshEntry> hipFvMeshes_;
static bool hipFvAnnounced_ = false;

static void hipFvCheck(int rc, const char* where)
{
    if (rc)
    {
        FatalErrorIn(where) << "libldugpu: " << ldu_last_error() << exit(FatalError);
    }
}

static void hipFvFree(hipFvMeshEntry& e)
{
    if (e.bnd) ldu_fv_boundary_destroy(e.bnd);
    if (e.addr) ldu_addr_destroy(e.addr);
    e.bnd = NULL;
    e.addr = NULL;
}

static hipFvMeshEntry& hipFvLookup(const fvMesh& mesh)
{
    const lduAddressing& la = mesh.lduAddr();
    const labelUList& l = la.lowerAddr();
    const labelUList& u = la.upperAddr();
    std::map<const fvMesh*, hipFvMeshEntry>::iterator it = hipFvMeshes_.find(&mesh);
    if
    (
        it != hipFvMeshes_.end()
     && (
            it->second.nCells != la.size() || it->second.nFaces != l.size()
         || it->second.lPtr != l.begin() || it->second.uPtr != u.begin()
         || (mesh.changing() && it->second.hash != hipFvHash(mesh, true))
        )
    )
    {
        hipFvFree(it->second);
        hipFvMeshes_.erase(it);
        it = hipFvMeshes_.end();
    }
    if (it == hipFvMeshes_.end())
    {
        hipFvMeshEntry e;
        e.addr = NULL;
        e.bnd = NULL;
        e.nCells = la.size();
        e.nFaces = l.size();
        e.lPtr = l.begin();
        e.uPtr = u.begin();
        e.hash = hipFvHash(mesh, true);
        ldu_ctx* ctx = hipLduSharedContext();
        hipFvCheck(ldu_addr_create(ctx, &e.addr, e.nCells, e.nFaces, l.begin(), u.begin()), "hipFvLookup(const fvMesh&)");
        hipFvCheck(ldu_addr_finalize(e.addr), "hipFvLookup(const fvMesh&)");
        const fvBoundaryMesh& bm = mesh.boundary();
        std::vector<int32_t> sizes(bm.size()), coupled(bm.size()), cells;
        forAll(bm, patchi)
        {
            const labelUList& fc = bm[patchi].faceCells();
            sizes[patchi] = fc.size();
            coupled[patchi] = bm[patchi].coupled() ? 1 : 0;
            forAll(fc, i) cells.push_back(fc[i]);
        }
        e.nPatchFaces = label(cells.size());
        int32_t none = 0;
        hipFvCheck
        (
            ldu_fv_boundary_create
            (
                e.addr, int32_t(bm.size()), sizes.size() ? &sizes[0] : &none, cells.size() ? &cells[0] : &none,
                coupled.size() ? &coupled[0] : &none, &e.bnd
            ),
            "hipFvLookup(const fvMesh&)"
        );
        it = hipFvMeshes_.insert(std::make_pair(&mesh, e)).first;
        if (!hipFvAnnounced_)
        {
            hipFvAnnounced_ = true;
            Info<< "[hipFvSchemes] finite-volume stencils on the device: " << e.nCells << " cells, " << e.nFaces
                << " internal faces, " << e.nPatchFaces << " patch faces in " << bm.size() << " patches" << endl;
        }
    }
    return it->second;
}

struct hipFvRegistryCleaner
{
    ~hipFvRegistryCleaner()
    {
        if (Pstream::parRun()) return;     // (see hipLduSolvers.C: no device calls after MPI_Finalize)
        for (std::map<const fvMesh*, hipFvMeshEntry>::iterator i = hipFvMeshes_.begin(); i != hipFvMeshes_.end(); ++i)
        {
            hipFvFree(i->second);
        }
        hipFvMeshes_.clear();
    }
};
static hipFvRegistryCleaner hipFvRegistryCleaner_;

// how often each stencil ran on the device (printed at exit with LDU_VERBOSE: the tests read it)
static label hipFvCalls_[3] = {0, 0, 0};
struct hipFvCounter
{
    ~hipFvCounter()
    {
        if (getenv("LDU_VERBOSE"))
        {
            Info<< "[hipFvSchemes] device calls: fvmLaplacian " << hipFvCalls_[0] << ", fvmDiv " << hipFvCalls_[1]
                << ", gaussGrad " << hipFvCalls_[2] << endl;
        }
    }
};
static hipFvCounter hipFvCounter_;


namespace fv
{

// ------------------------------------------------------------------ laplacianSchemes: hipGauss <interpolation> <snGrad>

// gaussLaplacianScheme<Type, scalar> with the internal-face part of fvmLaplacianUncorrected (gaussLaplacianScheme.C:46-88:
// upper = deltaCoeffs*gammaMagSf; negSumDiag) on the device.  The scalar-gamma flow around it is
// gaussLaplacianSchemes.C:43-96 (declareFvmLaplacianScalarGamma), restated because that specialisation is not virtual
// below fvmLaplacian.
template<class Type, class GType>
class hipGaussLaplacianScheme
:
    public gaussLaplacianScheme<Type, GType>
{
    hipGaussLaplacianScheme(const hipGaussLaplacianScheme&);
    void operator=(const hipGaussLaplacianScheme&);

public:

    TypeName("hipGauss");

    hipGaussLaplacianScheme(const fvMesh& mesh, Istream& is)
    :
        gaussLaplacianScheme<Type, GType>(mesh, is)
    {}

    virtual ~hipGaussLaplacianScheme()
    {}

    tmp<fvMatrix<Type> > fvmLaplacian
    (
        const GeometricField<GType, fvsPatchField, surfaceMesh>& gamma,
        const GeometricField<Type, fvPatchField, volMesh>& vf
    )
    {
        const fvMesh& mesh = this->mesh();
        hipFvMeshEntry& E = hipFvLookup(mesh);

        GeometricField<scalar, fvsPatchField, surfaceMesh> gammaMagSf(gamma*mesh.magSf());
        tmp<surfaceScalarField> tdeltaCoeffs = this->tsnGradScheme_().deltaCoeffs(vf);
        const surfaceScalarField& deltaCoeffs = tdeltaCoeffs();

        tmp<fvMatrix<Type> > tfvm
        (
            new fvMatrix<Type>(vf, deltaCoeffs.dimensions()*gammaMagSf.dimensions()*vf.dimensions())
        );
        fvMatrix<Type>& fvm = tfvm();

        // ---- the face loops, on the device
        scalarField& upper = fvm.upper();
        scalarField& diag = fvm.diag();
        hipFvCheck
        (
            ldu_fvm_laplacian
            (
                E.addr, deltaCoeffs.internalField().begin(), gammaMagSf.internalField().begin(), diag.begin(),
                upper.begin()
            ),
            "hipGaussLaplacianScheme::fvmLaplacian"
        );
        hipFvCalls_[0]++;

        forAll(vf.boundaryField(), patchi)
        {
            const fvPatchField<Type>& pvf = vf.boundaryField()[patchi];
            const fvsPatchScalarField& pGamma = gammaMagSf.boundaryField()[patchi];
            const fvsPatchScalarField& pDeltaCoeffs = deltaCoeffs.boundaryField()[patchi];

            if (pvf.coupled())
            {
                fvm.internalCoeffs()[patchi] = pGamma*pvf.gradientInternalCoeffs(pDeltaCoeffs);
                fvm.boundaryCoeffs()[patchi] = -pGamma*pvf.gradientBoundaryCoeffs(pDeltaCoeffs);
            }
            else
            {
                fvm.internalCoeffs()[patchi] = pGamma*pvf.gradientInternalCoeffs();
                fvm.boundaryCoeffs()[patchi] = -pGamma*pvf.gradientBoundaryCoeffs();
            }
        }

        // ---- the explicit non-orthogonal correction (gaussLaplacianSchemes.C:62-90), the reference's own operators
        if (this->tsnGradScheme_().corrected())
        {
            if (mesh.fluxRequired(vf.name()))
            {
                fvm.faceFluxCorrectionPtr() = new GeometricField<Type, fvsPatchField, surfaceMesh>
                (
                    gammaMagSf*this->tsnGradScheme_().correction(vf)
                );
                fvm.source() -= mesh.V()*fvc::div(*fvm.faceFluxCorrectionPtr())().internalField();
            }
            else
            {
                fvm.source() -=
                    mesh.V()*fvc::div(gammaMagSf*this->tsnGradScheme_().correction(vf))().internalField();
            }
        }
        return tfvm;
    }
};


// ------------------------------------------------------------------ divSchemes: [bounded] hipGauss <interpolation>

// gaussConvectionScheme<Type> with the coefficient loops of fvmDiv (gaussConvectionScheme.C:68-107: lower = -w*phi; upper =
// lower + phi; negSumDiag) on the device; weights, boundary coefficients and the explicit correction of a corrected
// interpolation scheme as in the reference.
template<class Type>
class hipGaussConvectionScheme
:
    public gaussConvectionScheme<Type>
{
    hipGaussConvectionScheme(const hipGaussConvectionScheme&);
    void operator=(const hipGaussConvectionScheme&);

public:

    TypeName("hipGauss");

    hipGaussConvectionScheme(const fvMesh& mesh, const surfaceScalarField& faceFlux, Istream& is)
    :
        gaussConvectionScheme<Type>(mesh, faceFlux, is)
    {}

    virtual ~hipGaussConvectionScheme()
    {}

    tmp<fvMatrix<Type> > fvmDiv
    (
        const surfaceScalarField& faceFlux,
        const GeometricField<Type, fvPatchField, volMesh>& vf
    ) const
    {
        hipFvMeshEntry& E = hipFvLookup(this->mesh());
        tmp<surfaceScalarField> tweights = this->tinterpScheme_().weights(vf);
        const surfaceScalarField& weights = tweights();

        tmp<fvMatrix<Type> > tfvm(new fvMatrix<Type>(vf, faceFlux.dimensions()*vf.dimensions()));
        fvMatrix<Type>& fvm = tfvm();

        scalarField& lower = fvm.lower();
        scalarField& upper = fvm.upper();
        scalarField& diag = fvm.diag();
        hipFvCheck
        (
            ldu_fvm_div
            (
                E.addr, weights.internalField().begin(), faceFlux.internalField().begin(), diag.begin(), upper.begin(),
                lower.begin()
            ),
            "hipGaussConvectionScheme::fvmDiv"
        );
        hipFvCalls_[1]++;

        forAll(vf.boundaryField(), patchI)
        {
            const fvPatchField<Type>& psf = vf.boundaryField()[patchI];
            const fvsPatchScalarField& patchFlux = faceFlux.boundaryField()[patchI];
            const fvsPatchScalarField& pw = weights.boundaryField()[patchI];

            fvm.internalCoeffs()[patchI] = patchFlux*psf.valueInternalCoeffs(pw);
            fvm.boundaryCoeffs()[patchI] = -patchFlux*psf.valueBoundaryCoeffs(pw);
        }

        if (this->tinterpScheme_().corrected())
        {
            fvm += fvc::surfaceIntegrate(faceFlux*this->tinterpScheme_().correction(vf));
        }
        return tfvm;
    }
};


// ------------------------------------------------------------------ gradSchemes: hipGauss <interpolation>

// gaussGrad<Type>::calcGrad (gaussGrad.C:112-141) with gradf's loops (gaussGrad.C:41-110: internal faces, patch faces,
// division by V) on the device; the face values come from the chosen interpolation scheme, the boundary conditions of the
// gradient field are corrected by the reference's own correctBoundaryConditions.
template<class Type>
class hipGaussGrad
:
    public gaussGrad<Type>
{
    hipGaussGrad(const hipGaussGrad&);
    void operator=(const hipGaussGrad&);

public:

    TypeName("hipGauss");

    hipGaussGrad(const fvMesh& mesh, Istream& is)
    :
        gaussGrad<Type>(mesh, is)
    {}

    virtual ~hipGaussGrad()
    {}

    virtual tmp<GeometricField<typename outerProduct<vector, Type>::type, fvPatchField, volMesh> > calcGrad
    (
        const GeometricField<Type, fvPatchField, volMesh>& vsf,
        const word& name
    ) const
    {
        typedef typename outerProduct<vector, Type>::type GradType;
        const fvMesh& mesh = vsf.mesh();
        hipFvMeshEntry& E = hipFvLookup(mesh);

        tmp<GeometricField<Type, fvsPatchField, surfaceMesh> > tssf = this->tinterpScheme_().interpolate(vsf);
        const GeometricField<Type, fvsPatchField, surfaceMesh>& ssf = tssf();

        tmp<GeometricField<GradType, fvPatchField, volMesh> > tgGrad
        (
            new GeometricField<GradType, fvPatchField, volMesh>
            (
                IOobject(name, ssf.instance(), mesh, IOobject::NO_READ, IOobject::NO_WRITE),
                mesh,
                dimensioned<GradType>("0", ssf.dimensions()/dimLength, pTraits<GradType>::zero),
                zeroGradientFvPatchField<GradType>::typeName
            )
        );
        GeometricField<GradType, fvPatchField, volMesh>& gGrad = tgGrad();

        // patch values concatenated in patch order
        Field<vector> bSf(E.nPatchFaces);
        Field<Type> bSsf(E.nPatchFaces);
        label k = 0;
        forAll(mesh.boundary(), patchi)
        {
            const vectorField& pSf = mesh.Sf().boundaryField()[patchi];
            const fvsPatchField<Type>& pssf = ssf.boundaryField()[patchi];
            forAll(mesh.boundary()[patchi], facei)
            {
                bSf[k] = pSf[facei];
                bSsf[k] = pssf[facei];
                k++;
            }
        }
        Field<GradType>& igGrad = gGrad;
        hipFvCheck
        (
            ldu_fvc_gaussGradFull
            (
                E.addr, E.bnd, int32_t(pTraits<Type>::nComponents),
                reinterpret_cast<const double*>(mesh.Sf().internalField().begin()),
                reinterpret_cast<const double*>(ssf.internalField().begin()),
                reinterpret_cast<const double*>(bSf.begin()),
                reinterpret_cast<const double*>(bSsf.begin()),
                mesh.V().field().begin(),
                reinterpret_cast<double*>(igGrad.begin())
            ),
            "hipGaussGrad::calcGrad"
        );
        hipFvCalls_[2]++;

        gGrad.correctBoundaryConditions();
        gaussGrad<Type>::correctBoundaryConditions(vsf, gGrad);
        return tgGrad;
    }
};

} // End namespace fv


// ------------------------------------------------------------------ run-time selection (the reference's own macros)

namespace fv
{
    makeFvLaplacianTypeScheme(hipGaussLaplacianScheme, scalar, scalar)
    makeFvLaplacianTypeScheme(hipGaussLaplacianScheme, scalar, vector)

    makeFvConvectionTypeScheme(hipGaussConvectionScheme, scalar)
    makeFvConvectionTypeScheme(hipGaussConvectionScheme, vector)

    makeFvGradTypeScheme(hipGaussGrad, scalar)
    makeFvGradTypeScheme(hipGaussGrad, vector)
}

} // End namespace Foam
