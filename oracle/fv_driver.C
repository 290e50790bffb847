// TEST INFRASTRUCTURE - runs the REFERENCE's own finite-volume stencils (libfiniteVolume units
// compiled by oracle/build_ref_fv.sh from /root/reference, linked statically) on a polyMesh case and
// dumps geometry + results as raw little-endian doubles, to pin oracle/fv_oracle.py and the HIP
// kernels ldu_fv_* / ldu_fvc_* / ldu_fvm_* (SURVEY.md 8a rows a30, a33-a39).
// Our code; only reference HEADERS are included.  Never shipped, never linked into the product.
//
// usage: fv_driver <caseDir> <in.bin> <out.bin>
//   in.bin : vf[nC] U[3 nC] phi[nF] gamma[nF]            (nF = internal faces)
//   out.bin: sections "name count" + doubles, see put()
#include "argList.H"
#include "Time.H"
#include "fvMesh.H"
#include "volFields.H"
#include "surfaceFields.H"
#include "linear.H"
#include "upwind.H"
#include "gaussGrad.H"
#include "snGradScheme.H"
#include "uncorrectedSnGrad.H"
#include "gaussLaplacianScheme.H"
#include "gaussConvectionScheme.H"
#include "fvcSurfaceIntegrate.H"
#include "fvMatrices.H"
#include "calculatedFvPatchFields.H"
#include "calculatedFvsPatchFields.H"
#include "zeroGradientFvPatchFields.H"
#include <cstdio>
#include <vector>

using namespace Foam;

static FILE* out = NULL;

static void put(const char* name, const double* p, long n)
{
    char hdr[64];
    memset(hdr, 0, sizeof(hdr));
    snprintf(hdr, sizeof(hdr), "%s %ld", name, n);
    fwrite(hdr, 1, sizeof(hdr), out);
    fwrite(p, sizeof(double), n, out);
}
static void put(const char* name, const scalarField& f) { put(name, f.begin(), f.size()); }
static void put(const char* name, const vectorField& f)
{
    put(name, reinterpret_cast<const double*>(f.begin()), 3L * f.size());
}

int main(int argc, char* argv[])
{
    if (argc != 4) { fprintf(stderr, "usage: fv_driver caseDir in.bin out.bin\n"); return 2; }
    fileName caseDir(argv[1]);
    Time runTime(Time::controlDictName, fileName(caseDir.path()), fileName(caseDir.name()));
    fvMesh mesh(IOobject(fvMesh::defaultRegion, runTime.timeName(), runTime, IOobject::MUST_READ));
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();

    std::vector<double> in((size_t)nC * 4 + (size_t)nF * 2);
    {
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(in.data(), sizeof(double), in.size(), f) != in.size())
        {
            fprintf(stderr, "fv_driver: short input (nC=%d nF=%d)\n", nC, nF);
            return 3;
        }
        fclose(f);
    }
    out = fopen(argv[3], "wb");

    // fields: internal values from the input, boundary values zero ("calculated")
    volScalarField vf(IOobject("vf", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0),
                      calculatedFvPatchScalarField::typeName);
    volVectorField U(IOobject("U", runTime.timeName(), mesh), mesh, dimensionedVector("0", dimless, vector::zero),
                     calculatedFvPatchVectorField::typeName);
    // (matrix assembly asks the patch fields for coefficients: zeroGradient has them, calculated does not)
    volScalarField vfz(IOobject("vfz", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0),
                       zeroGradientFvPatchScalarField::typeName);
    surfaceScalarField phi(IOobject("phi", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    surfaceScalarField gamma(IOobject("gamma", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    for (label c = 0; c < nC; c++)
    {
        vf.internalField()[c] = in[c];
        vfz.internalField()[c] = in[c];
        U.internalField()[c] = vector(in[nC + 3 * c], in[nC + 3 * c + 1], in[nC + 3 * c + 2]);
    }
    for (label f = 0; f < nF; f++)
    {
        phi.internalField()[f] = in[(size_t)4 * nC + f];
        gamma.internalField()[f] = in[(size_t)4 * nC + nF + f];
    }

    // ---- geometry (surfaceInterpolation.C:175-185, 239-242; fvMeshGeometry.C)
    put("weights", mesh.weights().internalField());
    put("deltaCoeffs", mesh.deltaCoeffs().internalField());
    put("nonOrthDeltaCoeffs", mesh.nonOrthDeltaCoeffs().internalField());
    put("V", mesh.V().field());
    put("Sf", mesh.Sf().internalField());
    put("magSf", mesh.magSf().internalField());
    {
        // faceAreaPairGAMGAgglomeration.C:48-73: mag(cmptMultiply(Sf/sqrt(magSf), vector(1, 1.01, 1.02)))
        scalarField w(mag(cmptMultiply(mesh.Sf().internalField() / sqrt(mesh.magSf().internalField()),
                                       vector(1, 1.01, 1.02))));
        put("faceAreaPairWeights", w);
    }

    // ---- a35 surfaceInterpolationScheme::interpolate with linear weights
    {
        linear<scalar> sch(mesh);
        tmp<surfaceScalarField> s = sch.interpolate(vf);
        put("interpolate_s", s().internalField());
        linear<vector> schv(mesh);
        tmp<surfaceVectorField> v = schv.interpolate(U);
        put("interpolate_v", v().internalField());
    }
    // ---- upwind weights pos(faceFlux) (upwind.H) and interpolation with them
    {
        upwind<scalar> sch(mesh, phi);
        tmp<surfaceScalarField> w = sch.weights(vf);
        put("upwindWeights", w().internalField());
        tmp<surfaceScalarField> s = sch.interpolate(vf);
        put("interpolate_upwind", s().internalField());
    }
    // ---- a33 fvc::surfaceIntegrate (boundary flux zero)
    {
        tmp<volScalarField> d = fvc::surfaceIntegrate(phi);
        put("surfaceIntegrate_s", d().internalField());
        surfaceVectorField phiU("phiU", linear<vector>(mesh).interpolate(U) * phi);
        forAll(phiU.boundaryField(), p) phiU.boundaryField()[p] = vector::zero;
        tmp<volVectorField> dv = fvc::surfaceIntegrate(phiU);
        put("surfaceIntegrate_v", dv().internalField());
        put("phiU", phiU.internalField());
    }
    // ---- a34 gaussGrad::gradf of a face field with zero boundary values
    {
        surfaceScalarField ssf("ssf", linear<scalar>(mesh).interpolate(vf));
        forAll(ssf.boundaryField(), p) ssf.boundaryField()[p] = 0.0;
        tmp<volVectorField> g = fv::gaussGrad<scalar>::gradf(ssf, "grad(vf)");
        put("gaussGrad", g().internalField());
    }
    // ---- a36 snGradScheme::snGrad with deltaCoeffs
    {
        tmp<surfaceScalarField> s = fv::snGradScheme<scalar>::snGrad(vf, mesh.deltaCoeffs(), "snGrad");
        put("snGrad", s().internalField());
    }
    // ---- a37 gaussLaplacianScheme::fvmLaplacianUncorrected
    {
        surfaceScalarField gammaMagSf("gammaMagSf", gamma * mesh.magSf());
        put("gammaMagSf", gammaMagSf.internalField());
        tmp<fvScalarMatrix> M = fv::gaussLaplacianScheme<scalar, scalar>::fvmLaplacianUncorrected(gammaMagSf, mesh.deltaCoeffs(), vfz);
        put("laplacian_upper", M().upper());
        put("laplacian_diag", M().diag());
    }
    // ---- a38 gaussConvectionScheme::fvmDiv with linear and with upwind weights
    {
        fv::gaussConvectionScheme<scalar> cs(mesh, phi,
            tmp<surfaceInterpolationScheme<scalar> >(new linear<scalar>(mesh)));
        tmp<fvScalarMatrix> M = cs.fvmDiv(phi, vfz);
        put("div_linear_lower", M().lower());
        put("div_linear_upper", M().upper());
        put("div_linear_diag", M().diag());
        fv::gaussConvectionScheme<scalar> cu(mesh, phi,
            tmp<surfaceInterpolationScheme<scalar> >(new upwind<scalar>(mesh, phi)));
        tmp<fvScalarMatrix> Mu = cu.fvmDiv(phi, vfz);
        put("div_upwind_lower", Mu().lower());
        put("div_upwind_upper", Mu().upper());
        put("div_upwind_diag", Mu().diag());
    }
    // addressing as the reference sees it (must equal the generator's)
    {
        scalarField l(nF), u(nF);
        forAll(l, f) { l[f] = mesh.owner()[f]; u[f] = mesh.neighbour()[f]; }
        put("owner", l);
        put("neighbour", u);
    }
    fclose(out);
    return 0;
}
