// TEST INFRASTRUCTURE - runs the REFERENCE's own finite-volume stencils (libfiniteVolume units
// compiled by oracle/build_ref_fv.sh from /root/reference, linked statically) on a polyMesh case and
// dumps geometry + results as raw little-endian doubles, to pin oracle/fv_oracle.py and the HIP
// kernels ldu_fv_* / ldu_fvc_* / ldu_fvm_* (SURVEY.md 8a rows a30, a33-a39).
// Our code; only reference HEADERS are included.  Never shipped, never linked into the product.
//
// usage: fv_driver <caseDir> <in.bin> <out.bin> [stencils|glue|glueV|solve|solve2] [extra GAMG controls for solve]
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
#include "correctedSnGrad.H"
#include "gaussDivScheme.H"
#include "fvcGrad.H"
#include "symmTensorField.H"
#include "gaussLaplacianScheme.H"
#include "gaussConvectionScheme.H"
#include "linearUpwind.H"
#include "fvmSup.H"
#include "fvcSurfaceIntegrate.H"
#include "linearUpwindV.H"
#include "cellLimitedGrad.H"
#include "fvcSurfaceIntegrate.H"
#include "fvMatrices.H"
#include "calculatedFvPatchFields.H"
#include "calculatedFvsPatchFields.H"
#include "zeroGradientFvPatchFields.H"
#include "cyclicFvPatch.H"
#include "cyclicFvPatchFields.H"
#include "cyclicFvsPatchFields.H"
#include "fixedValueFvPatchFields.H"
#include <cstdio>
#include <string>
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
static void put(const char* name, const tensorField& f)
{
    put(name, reinterpret_cast<const double*>(f.begin()), 9L * f.size());
}
static void put(const char* name, const symmTensorField& f)
{
    put(name, reinterpret_cast<const double*>(f.begin()), 6L * f.size());
}
static std::string pname(label p, const char* what)
{
    char nm[64];
    snprintf(nm, sizeof(nm), "p%d_%s", int(p), what);
    return std::string(nm);
}

// fvMatrix glue: a scalar transport matrix with fixedValue / zeroGradient / cyclic
// patches; dumps the matrix, the patch coefficient sets and what the reference's own fvMatrix methods
// make of them (fvMatrix.C: addBoundaryDiag, addBoundarySource, A, H, flux, relax, setReference).
static int glue(fvMesh& mesh, Time& runTime, const std::vector<double>& in)
{
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();
    dimensionSet::debug = 0;   // synthetic dimensionless fields: no dimension checking of the matrix sum
    wordList types(mesh.boundary().size());
    forAll(types, p)
    {
        types[p] = mesh.boundary()[p].coupled() ? word("cyclic") : (p % 2 ? word("zeroGradient") : word("fixedValue"));
    }
    volScalarField T(IOobject("T", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0), types);
    for (label c = 0; c < nC; c++) T.internalField()[c] = in[c];
    forAll(types, p)
    {
        if (types[p] == "fixedValue") T.boundaryField()[p] == scalar(0.7 + 0.1 * p);
    }
    T.correctBoundaryConditions();
    surfaceScalarField phi(IOobject("phi", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    surfaceScalarField gamma(IOobject("gamma", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    for (label f = 0; f < nF; f++)
    {
        phi.internalField()[f] = in[(size_t)4 * nC + f];
        gamma.internalField()[f] = in[(size_t)4 * nC + nF + f];
    }
    forAll(phi.boundaryField(), p)
    {
        forAll(phi.boundaryField()[p], i)
        {
            phi.boundaryField()[p][i] = 0.05 * ((i % 5) - 2) * (mesh.boundary()[p].coupled() ? 0.0 : 1.0);
            gamma.boundaryField()[p][i] = 0.8 + 0.01 * (i % 7);
        }
    }
    surfaceScalarField gammaMagSf("gammaMagSf", gamma * mesh.magSf());
    tmp<fvScalarMatrix> tLap =
        fv::gaussLaplacianScheme<scalar, scalar>::fvmLaplacianUncorrected(gammaMagSf, mesh.deltaCoeffs(), T);
    fv::gaussConvectionScheme<scalar> cs(mesh, phi, tmp<surfaceInterpolationScheme<scalar> >(new upwind<scalar>(mesh, phi)));
    tmp<fvScalarMatrix> tDiv = cs.fvmDiv(phi, T);
    fvScalarMatrix M(tDiv() - tLap());
    for (label c = 0; c < nC; c++) M.source()[c] = 0.3 * in[nC + 3 * c];

    put("diag", M.diag());
    put("upper", M.upper());
    put("lower", M.lower());
    put("source", M.source());
    put("psi", T.internalField());
    put("V", mesh.V().field());
    {
        scalarField np(1, scalar(mesh.boundary().size()));
        put("nPatches", np);
    }
    forAll(mesh.boundary(), p)
    {
        const labelUList& fc = mesh.lduAddr().patchAddr(p);
        scalarField fcd(fc.size());
        forAll(fc, i) fcd[i] = fc[i];
        char nm[64];
        snprintf(nm, sizeof(nm), "p%d_faceCells", p); put(nm, fcd);
        scalarField cp(1, T.boundaryField()[p].coupled() ? 1.0 : 0.0);
        snprintf(nm, sizeof(nm), "p%d_coupled", p); put(nm, cp);
        snprintf(nm, sizeof(nm), "p%d_internalCoeffs", p); put(nm, M.internalCoeffs()[p]);
        snprintf(nm, sizeof(nm), "p%d_boundaryCoeffs", p); put(nm, M.boundaryCoeffs()[p]);
        scalarField pnf(fc.size(), 0.0);
        if (T.boundaryField()[p].coupled()) pnf = T.boundaryField()[p].patchNeighbourField();
        snprintf(nm, sizeof(nm), "p%d_pnf", p); put(nm, pnf);
    }
    {
        scalarField d(M.diag());
        M.addBoundaryDiag(d, 0);
        put("ref_addBoundaryDiag", d);
        scalarField s(M.source());
        M.addBoundarySource(s);
        put("ref_addBoundarySource", s);
        scalarField s2(M.source());
        M.addBoundarySource(s2, false);
        put("ref_addBoundarySource_nocouples", s2);
    }
    put("ref_A", M.A()().internalField());
    put("ref_H", M.H()().internalField());
    {
        tmp<surfaceScalarField> fl = M.flux();
        put("ref_flux_internal", fl().internalField());
        forAll(fl().boundaryField(), p)
        {
            char nm[64];
            snprintf(nm, sizeof(nm), "p%d_ref_flux", p);
            scalarField b(fl().boundaryField()[p]);
            put(nm, b);
        }
    }
    {
        fvScalarMatrix R(M);
        R.relax(0.7);
        put("ref_relax_diag", R.diag());
        put("ref_relax_source", R.source());
    }
    {
        fvScalarMatrix R(M);
        R.setReference(5, 1.3, true);
        put("ref_setReference_diag", R.diag());
        put("ref_setReference_source", R.source());
    }
    fclose(out);
    return 0;
}

// the same glue for a VECTOR matrix (U-equation like): fvm::div(phi, U) - fvm::laplacian(gamma, U) with
// fixedValue / zeroGradient / cyclic patches
static int glueV(fvMesh& mesh, Time& runTime, const std::vector<double>& in)
{
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();
    dimensionSet::debug = 0;
    wordList types(mesh.boundary().size());
    forAll(types, p)
    {
        types[p] = mesh.boundary()[p].coupled() ? word("cyclic") : (p % 2 ? word("zeroGradient") : word("fixedValue"));
    }
    volVectorField U(IOobject("U", runTime.timeName(), mesh), mesh, dimensionedVector("0", dimless, vector::zero), types);
    for (label c = 0; c < nC; c++) U.internalField()[c] = vector(in[nC + 3 * c], in[nC + 3 * c + 1], in[nC + 3 * c + 2]);
    forAll(types, p)
        if (types[p] == "fixedValue") U.boundaryField()[p] == vector(0.3 + 0.1 * p, -0.2, 0.05 * p);
    U.correctBoundaryConditions();
    surfaceScalarField phi(IOobject("phi", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    surfaceScalarField gamma(IOobject("gamma", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    for (label f = 0; f < nF; f++)
    {
        phi.internalField()[f] = in[(size_t)4 * nC + f];
        gamma.internalField()[f] = in[(size_t)4 * nC + nF + f];
    }
    forAll(phi.boundaryField(), p)
        forAll(phi.boundaryField()[p], i)
        {
            phi.boundaryField()[p][i] = 0.05 * ((i % 5) - 2) * (mesh.boundary()[p].coupled() ? 0.0 : 1.0);
            gamma.boundaryField()[p][i] = 0.8 + 0.01 * (i % 7);
        }
    surfaceScalarField gammaMagSf("gammaMagSf", gamma * mesh.magSf());
    tmp<fvVectorMatrix> tLap =
        fv::gaussLaplacianScheme<vector, scalar>::fvmLaplacianUncorrected(gammaMagSf, mesh.deltaCoeffs(), U);
    fv::gaussConvectionScheme<vector> cs(mesh, phi, tmp<surfaceInterpolationScheme<vector> >(new upwind<vector>(mesh, phi)));
    tmp<fvVectorMatrix> tDiv = cs.fvmDiv(phi, U);
    fvVectorMatrix M(tDiv() - tLap());
    for (label c = 0; c < nC; c++) M.source()[c] = 0.3 * vector(in[c], in[nC + 3 * c], -in[c]);

    put("diag", M.diag());
    put("upper", M.upper());
    put("lower", M.lower());
    put("source", M.source());
    put("psi", U.internalField());
    put("V", mesh.V().field());
    {
        scalarField np(1, scalar(mesh.boundary().size()));
        put("nPatches", np);
    }
    forAll(mesh.boundary(), p)
    {
        const labelUList& fc = mesh.lduAddr().patchAddr(p);
        scalarField fcd(fc.size());
        forAll(fc, i) fcd[i] = fc[i];
        char nm[64];
        snprintf(nm, sizeof(nm), "p%d_faceCells", p); put(nm, fcd);
        scalarField cp(1, U.boundaryField()[p].coupled() ? 1.0 : 0.0);
        snprintf(nm, sizeof(nm), "p%d_coupled", p); put(nm, cp);
        snprintf(nm, sizeof(nm), "p%d_internalCoeffs", p); put(nm, M.internalCoeffs()[p]);
        snprintf(nm, sizeof(nm), "p%d_boundaryCoeffs", p); put(nm, M.boundaryCoeffs()[p]);
        vectorField pnf(fc.size(), vector::zero);
        if (U.boundaryField()[p].coupled()) pnf = U.boundaryField()[p].patchNeighbourField();
        snprintf(nm, sizeof(nm), "p%d_pnf", p); put(nm, pnf);
    }
    for (direction cmpt = 0; cmpt < 3; cmpt++)
    {
        scalarField d(M.diag());
        M.addBoundaryDiag(d, cmpt);
        char nm[64];
        snprintf(nm, sizeof(nm), "ref_addBoundaryDiag%d", cmpt); put(nm, d);
    }
    {
        vectorField s(M.source());
        M.addBoundarySource(s);
        put("ref_addBoundarySource", s);
        vectorField s2(M.source());
        M.addBoundarySource(s2, false);
        put("ref_addBoundarySource_nocouples", s2);
    }
    put("ref_A", M.A()().internalField());
    put("ref_H", M.H()().internalField());
    {
        fvVectorMatrix R(M);
        R.relax(0.7);
        put("ref_relax_diag", R.diag());
        put("ref_relax_source", R.source());
    }
    // `type coupled;`: fvMatrix<vector>::solve -> solveCoupled (fvMatrixSolve.C:83-85, :222-277) builds an
    // LduMatrix<vector, scalar, scalar> with the component-0 boundary coefficients on the cyclic interfaces
    // and runs the templated solvers.  solveCoupled returns an empty solverPerformance: psi is the evidence.
    {
        const vectorField U0(U.internalField());
        const char* names[3] = {"PBiCCCG", "PBiCICG", "SmoothSolver"};
        for (int k = 0; k < 3; k++)
        {
            U.internalField() = U0;
            U.correctBoundaryConditions();
            fvVectorMatrix S(M);
            std::string ds = std::string("type coupled; solver ") + names[k] + "; preconditioner DILU; "
                "smoother GaussSeidel; nSweeps 2; tolerance (1e-9 1e-9 1e-9); relTol (0 0 0); maxIter 40;";
            dictionary d(IStringStream(ds.c_str())());
            S.solve(d);
            put((std::string("ref_coupled_") + names[k]).c_str(), U.internalField());
        }
        // the default (segregated) path of a vector equation: three scalar solves with the component's
        // boundary coefficients (fvMatrixSolve.C:103-218)
        {
            U.internalField() = U0;
            U.correctBoundaryConditions();
            fvVectorMatrix S(M);
            dictionary d(IStringStream("solver PBiCG; preconditioner DILU; tolerance 1e-10; relTol 0; maxIter 60;")());
            S.solve(d);
            put("ref_segregated_PBiCG", U.internalField());
        }
        U.internalField() = U0;
        U.correctBoundaryConditions();
    }
    fclose(out);
    return 0;
}

// End to end through the reference's own application-level call: fvScalarMatrix::solve(dict) ->
// solveSegregated (fvScalarMatrix.C:136-183: addBoundaryDiag, addBoundarySource(couples=false),
// lduMatrix::solver::New(...)->solve) with GAMG + the REAL faceAreaPairGAMGAgglomeration of
// libfiniteVolume (weights from mesh.Sf()), and with PCG/DIC.
static int solveMode(fvMesh& mesh, Time& runTime, const std::vector<double>& in, const char* extraControls)
{
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();
    dimensionSet::debug = 0;
    wordList types(mesh.boundary().size());
    forAll(types, p)
    {
        types[p] = mesh.boundary()[p].coupled() ? word("cyclic") : (p % 2 ? word("zeroGradient") : word("fixedValue"));
    }
    surfaceScalarField gamma(IOobject("gamma", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    for (label f = 0; f < nF; f++) gamma.internalField()[f] = in[(size_t)4 * nC + nF + f];
    forAll(gamma.boundaryField(), p)
        forAll(gamma.boundaryField()[p], i) gamma.boundaryField()[p][i] = 0.8 + 0.01 * (i % 7);
    surfaceScalarField gammaMagSf("gammaMagSf", gamma * mesh.magSf());
    {
        scalarField w(mag(cmptMultiply(mesh.Sf().internalField() / sqrt(mesh.magSf().internalField()),
                                       vector(1, 1.01, 1.02))));
        put("faceAreaPairWeights", w);
    }
    const bool ownSmoother = std::string(extraControls).find("smoother ") != std::string::npos;
    std::string d0 = std::string("solver GAMG; ") + (ownSmoother ? "" : "smoother GaussSeidel; ")
        + std::string("agglomerator faceAreaPair; mergeLevels 1; "
        "cacheAgglomeration off; tolerance 1e-10; relTol 0; nPreSweeps 0; nPostSweeps 2; nFinestSweeps 2; ")
        + extraControls;
    // "asym" in the controls: add an upwind convection term (asymmetric matrix, interface coefficients
    // that differ between the two sides) and use PBiCG/DILU as the Krylov solver
    const bool asym = std::string(extraControls).find("asymmetric") != std::string::npos;
    if (asym) d0 = d0.substr(0, d0.find("asymmetric"));
    const char* dicts[2] = {d0.c_str(), asym ? "solver PBiCG; preconditioner DILU; tolerance 1e-10; relTol 0;"
                                             : "solver PCG; preconditioner DIC; tolerance 1e-10; relTol 0;"};
    surfaceScalarField phi(IOobject("phi", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    if (asym)
    {
        for (label f = 0; f < nF; f++) phi.internalField()[f] = 0.004 * in[(size_t)4 * nC + f];
        forAll(phi.boundaryField(), p)
        {
            if (!mesh.boundary()[p].coupled()) continue;
            // the flux through a cyclic face leaves one side and enters the other
            const scalar sgn = (p % 2) ? -1.0 : 1.0;
            forAll(phi.boundaryField()[p], i) phi.boundaryField()[p][i] = sgn * 0.003 * (1 + (i % 4));
        }
    }
    for (int k = 0; k < 2; k++)
    {
        volScalarField T(IOobject(k ? "Tp" : "Tg", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0),
                         types);
        for (label c = 0; c < nC; c++) T.internalField()[c] = 0.0;
        forAll(types, p)
            if (types[p] == "fixedValue") T.boundaryField()[p] == scalar(0.7 + 0.1 * p);
        tmp<fvScalarMatrix> tLap =
            fv::gaussLaplacianScheme<scalar, scalar>::fvmLaplacianUncorrected(gammaMagSf, mesh.deltaCoeffs(), T);
        fvScalarMatrix M(-tLap());
        if (asym)
        {
            fv::gaussConvectionScheme<scalar> cs(mesh, phi,
                tmp<surfaceInterpolationScheme<scalar> >(new upwind<scalar>(mesh, phi)));
            M += cs.fvmDiv(phi, T);
        }
        for (label c = 0; c < nC; c++) M.source()[c] = 0.01 * in[c];
        if (k == 0)
        {
            put("diag", M.diag());
            put("upper", M.upper());
            if (asym) put("lower", M.lower());
            put("source", M.source());
            scalarField np(1, scalar(mesh.boundary().size()));
            put("nPatches", np);
            forAll(mesh.boundary(), p)
            {
                const labelUList& fc = mesh.lduAddr().patchAddr(p);
                scalarField fcd(fc.size());
                forAll(fc, i) fcd[i] = fc[i];
                char nm[64];
                snprintf(nm, sizeof(nm), "p%d_faceCells", p); put(nm, fcd);
                snprintf(nm, sizeof(nm), "p%d_internalCoeffs", p); put(nm, M.internalCoeffs()[p]);
                snprintf(nm, sizeof(nm), "p%d_boundaryCoeffs", p); put(nm, M.boundaryCoeffs()[p]);
                scalarField cp(1, T.boundaryField()[p].coupled() ? 1.0 : 0.0);
                snprintf(nm, sizeof(nm), "p%d_coupled", p); put(nm, cp);
            }
            // the smoothers themselves on this system (what solveSegregated hands over), 3 sweeps from a
            // non-trivial start: bit-level reference for GaussSeidel and nonBlockingGaussSeidel with
            // coupled (cyclic) interfaces
            scalarField saveDiag(M.diag());
            M.addBoundaryDiag(M.diag(), 0);
            scalarField totalSource(M.source());
            M.addBoundarySource(totalSource, false);
            const char* smNames[2] = {"GaussSeidel", "nonBlockingGaussSeidel"};
            scalarField x0(nC);
            for (label c = 0; c < nC; c++) x0[c] = 0.1 * in[nC + 3 * c + 1];
            put("smooth_x0", x0);
            for (int si = 0; si < 2; si++)
            {
                dictionary sd;
                sd.add("smoother", word(smNames[si]));
                scalarField x(x0);
                lduMatrix::smoother::New(T.name(), M, M.boundaryCoeffs(), M.internalCoeffs(),
                                         T.boundaryField().scalarInterfaces(), sd)->smooth(x, totalSource, 0, 3);
                put((std::string("ref_smooth_") + smNames[si]).c_str(), x);
            }
            M.diag() = saveDiag;
        }
        dictionary d(IStringStream(dicts[k])());
        solverPerformance perf = M.solve(d);
        scalarField pf(4);
        pf[0] = perf.initialResidual(); pf[1] = perf.finalResidual(); pf[2] = perf.nIterations(); pf[3] = perf.converged();
        put(k ? "ref_pcg_perf" : "ref_gamg_perf", pf);
        put(k ? "ref_pcg_psi" : "ref_gamg_psi", T.internalField());
    }
    fclose(out);
    return 0;
}

// SURVEY.md 8a rows a36 / a37 / a39 and the patch halves of a34 / a35 on a non-orthogonal mesh:
// nonOrthDeltaCoeffs / nonOrthCorrectionVectors (surfaceInterpolation.C:252-391), correctedSnGrad (correction, full
// snGrad), the corrected gaussLaplacianScheme with scalar / symmTensor / tensor gamma (fvm and fvc), gaussDivScheme of
// a vector and of a tensor field, interpolation on the patch faces (coupled and not) and gaussGrad's boundary
// correction - all computed by the reference's own classes; fixedValue / zeroGradient / cyclic patches.
static int nonorth(fvMesh& mesh, Time& runTime, const std::vector<double>& in)
{
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();
    dimensionSet::debug = 0;
    wordList types(mesh.boundary().size());
    forAll(types, p)
    {
        types[p] = mesh.boundary()[p].coupled() ? word("cyclic") : (p % 2 ? word("zeroGradient") : word("fixedValue"));
    }
    volScalarField T(IOobject("T", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0), types);
    volVectorField U(IOobject("U", runTime.timeName(), mesh), mesh, dimensionedVector("0", dimless, vector::zero), types);
    for (label c = 0; c < nC; c++)
    {
        T.internalField()[c] = in[c];
        U.internalField()[c] = vector(in[nC + 3 * c], in[nC + 3 * c + 1], in[nC + 3 * c + 2]);
    }
    forAll(types, p)
    {
        if (types[p] == "fixedValue")
        {
            scalarField tv(mesh.boundary()[p].size());
            vectorField uv(mesh.boundary()[p].size());
            forAll(tv, i)
            {
                tv[i] = 0.7 + 0.1 * p + 0.03 * (i % 5);
                uv[i] = vector(0.3 + 0.1 * p, -0.2 + 0.02 * (i % 3), 0.05 * p);
            }
            T.boundaryField()[p] == tv;
            U.boundaryField()[p] == uv;
        }
    }
    T.correctBoundaryConditions();
    U.correctBoundaryConditions();
    surfaceScalarField gamma(IOobject("gamma", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0));
    for (label f = 0; f < nF; f++) gamma.internalField()[f] = in[(size_t)4 * nC + nF + f];
    forAll(gamma.boundaryField(), p)
        forAll(gamma.boundaryField()[p], i) gamma.boundaryField()[p][i] = 0.8 + 0.01 * (i % 7);
    // cyclic halves must carry the same face value
    forAll(gamma.boundaryField(), p)
        if (mesh.boundary()[p].coupled())
            forAll(gamma.boundaryField()[p], i) gamma.boundaryField()[p][i] = 0.9 + 0.02 * (i % 4);

    // ---- geometry
    put("V", mesh.V().field());
    put("C", mesh.C().internalField());
    put("Sf", mesh.Sf().internalField());
    put("magSf", mesh.magSf().internalField());
    put("weights", mesh.weights().internalField());
    put("deltaCoeffs", mesh.deltaCoeffs().internalField());
    put("nonOrthDeltaCoeffs", mesh.nonOrthDeltaCoeffs().internalField());
    put("nonOrthCorrectionVectors", mesh.nonOrthCorrectionVectors().internalField());
    put("T", T.internalField());
    put("U", U.internalField());
    put("gamma", gamma.internalField());
    {
        scalarField np(1, scalar(mesh.boundary().size()));
        put("nPatches", np);
    }
    forAll(mesh.boundary(), p)
    {
        const fvPatch& fp = mesh.boundary()[p];
        const labelUList& fc = fp.faceCells();
        scalarField fcd(fc.size());
        forAll(fc, i) fcd[i] = fc[i];
        put(pname(p, "faceCells").c_str(), fcd);
        scalarField cp(1, fp.coupled() ? 1.0 : 0.0);
        put(pname(p, "coupled").c_str(), cp);
        put(pname(p, "Sf").c_str(), mesh.Sf().boundaryField()[p]);
        put(pname(p, "magSf").c_str(), mesh.magSf().boundaryField()[p]);
        vectorField d(fp.delta());
        put(pname(p, "delta").c_str(), d);
        vectorField nf(fp.nf());
        put(pname(p, "nf").c_str(), nf);
        put(pname(p, "weights").c_str(), mesh.weights().boundaryField()[p]);
        put(pname(p, "deltaCoeffs").c_str(), mesh.deltaCoeffs().boundaryField()[p]);
        put(pname(p, "nonOrthDeltaCoeffs").c_str(), mesh.nonOrthDeltaCoeffs().boundaryField()[p]);
        put(pname(p, "nonOrthCorrectionVectors").c_str(), mesh.nonOrthCorrectionVectors().boundaryField()[p]);
        put(pname(p, "gamma").c_str(), gamma.boundaryField()[p]);
        put(pname(p, "T").c_str(), T.boundaryField()[p]);
        put(pname(p, "U").c_str(), U.boundaryField()[p]);
        scalarField tpnf(fc.size(), 0.0);
        vectorField upnf(fc.size(), vector::zero);
        if (fp.coupled())
        {
            tpnf = T.boundaryField()[p].patchNeighbourField();
            upnf = U.boundaryField()[p].patchNeighbourField();
        }
        put(pname(p, "T_pnf").c_str(), tpnf);
        put(pname(p, "U_pnf").c_str(), upnf);
        // fvPatchField::snGrad() = deltaCoeffs*(*this - patchInternalField()) (coupled patches have none without
        // an explicit deltaCoeffs argument, coupledFvPatchField.H:150)
        scalarField tsg(fc.size(), 0.0);
        vectorField usg(fc.size(), vector::zero);
        if (!fp.coupled())
        {
            tsg = T.boundaryField()[p].snGrad();
            usg = U.boundaryField()[p].snGrad();
        }
        put(pname(p, "T_snGrad").c_str(), tsg);
        put(pname(p, "U_snGrad").c_str(), usg);
    }

    // ---- a35 patch half: surfaceInterpolationScheme::interpolate on the patch faces (surfaceInterpolationScheme.C:298-314)
    {
        tmp<surfaceScalarField> s = linear<scalar>(mesh).interpolate(T);
        tmp<surfaceVectorField> v = linear<vector>(mesh).interpolate(U);
        put("ref_interpolate_T", s().internalField());
        put("ref_interpolate_U", v().internalField());
        forAll(mesh.boundary(), p)
        {
            put(pname(p, "ref_interpolate_T").c_str(), s().boundaryField()[p]);
            put(pname(p, "ref_interpolate_U").c_str(), v().boundaryField()[p]);
        }
    }
    // ---- a34 with its boundary correction (gaussGrad.C:123-170): Gauss linear gradients, internal and patch fields
    volVectorField gradT(fvc::grad(T));
    volTensorField gradU(fvc::grad(U));
    put("ref_gradT", gradT.internalField());
    put("ref_gradU", gradU.internalField());
    forAll(mesh.boundary(), p)
    {
        put(pname(p, "ref_gradT").c_str(), gradT.boundaryField()[p]);
        put(pname(p, "ref_gradU").c_str(), gradU.boundaryField()[p]);
        vectorField gpnf(mesh.boundary()[p].size(), vector::zero);
        tensorField gupnf(mesh.boundary()[p].size(), tensor::zero);
        if (mesh.boundary()[p].coupled())
        {
            gpnf = gradT.boundaryField()[p].patchNeighbourField();
            gupnf = gradU.boundaryField()[p].patchNeighbourField();
        }
        put(pname(p, "gradT_pnf").c_str(), gpnf);
        put(pname(p, "gradU_pnf").c_str(), gupnf);
    }
    // ---- a36: correctedSnGrad (correctedSnGrad.C:44-107, correctedSnGrads.C), snGradScheme::snGrad (snGradScheme.C:104-186)
    {
        fv::correctedSnGrad<scalar> cs(mesh);
        fv::correctedSnGrad<vector> cv(mesh);
        tmp<surfaceScalarField> c1 = cs.correction(T);
        tmp<surfaceVectorField> c3 = cv.correction(U);
        tmp<surfaceScalarField> g1 = cs.snGrad(T);
        tmp<surfaceVectorField> g3 = cv.snGrad(U);
        put("ref_snGradCorrection_T", c1().internalField());
        put("ref_snGradCorrection_U", c3().internalField());
        put("ref_correctedSnGrad_T", g1().internalField());
        put("ref_correctedSnGrad_U", g3().internalField());
        forAll(mesh.boundary(), p)
        {
            put(pname(p, "ref_snGradCorrection_T").c_str(), c1().boundaryField()[p]);
            put(pname(p, "ref_snGradCorrection_U").c_str(), c3().boundaryField()[p]);
        }
    }
    // ---- a37: gaussLaplacianScheme with the corrected snGrad, scalar gamma (gaussLaplacianSchemes.C:43-114)
    {
        fv::gaussLaplacianScheme<scalar, scalar> ls
        (
            mesh, tmp<surfaceInterpolationScheme<scalar> >(new linear<scalar>(mesh)),
            tmp<fv::snGradScheme<scalar> >(new fv::correctedSnGrad<scalar>(mesh))
        );
        tmp<fvScalarMatrix> M = ls.fvmLaplacian(gamma, T);
        put("ref_lap_upper", M().upper());
        put("ref_lap_diag", M().diag());
        put("ref_lap_source", M().source());
        forAll(mesh.boundary(), p)
        {
            put(pname(p, "ref_lap_internalCoeffs").c_str(), M().internalCoeffs()[p]);
            put(pname(p, "ref_lap_boundaryCoeffs").c_str(), M().boundaryCoeffs()[p]);
        }
        if (M().faceFluxCorrectionPtr())
        {
            put("ref_lap_faceFluxCorrection", M().faceFluxCorrectionPtr()->internalField());
            forAll(mesh.boundary(), p)
                put(pname(p, "ref_lap_faceFluxCorrection").c_str(), M().faceFluxCorrectionPtr()->boundaryField()[p]);
        }
        tmp<volScalarField> L = ls.fvcLaplacian(gamma, T);
        put("ref_fvcLaplacian", L().internalField());

        fv::gaussLaplacianScheme<vector, scalar> lv
        (
            mesh, tmp<surfaceInterpolationScheme<scalar> >(new linear<scalar>(mesh)),
            tmp<fv::snGradScheme<vector> >(new fv::correctedSnGrad<vector>(mesh))
        );
        tmp<fvVectorMatrix> MV = lv.fvmLaplacian(gamma, U);
        put("ref_lapU_upper", MV().upper());
        put("ref_lapU_diag", MV().diag());
        put("ref_lapU_source", MV().source());
    }
    // ---- a37: tensor-gamma path (gaussLaplacianScheme.C:92-231), symmTensor and tensor diffusivities
    {
        surfaceSymmTensorField gS
        (
            IOobject("gS", runTime.timeName(), mesh), mesh, dimensionedSymmTensor("0", dimless, symmTensor::zero)
        );
        surfaceTensorField gT
        (
            IOobject("gT", runTime.timeName(), mesh), mesh, dimensionedTensor("0", dimless, tensor::zero)
        );
        for (label f = 0; f < nF; f++)
        {
            const scalar g = gamma.internalField()[f];
            const scalar a = 0.1 * in[(size_t)4 * nC + f];
            gS.internalField()[f] = symmTensor(g, 0.1 * a, -0.05 * a, 1.2 * g, 0.07 * a, 0.9 * g);
            gT.internalField()[f] = tensor(g, 0.1 * a, -0.05 * a, 0.02 * a, 1.2 * g, 0.07 * a, 0.03 * a, -0.04 * a, 0.9 * g);
        }
        forAll(gS.boundaryField(), p)
            forAll(gS.boundaryField()[p], i)
            {
                const scalar g = gamma.boundaryField()[p][i];
                const scalar a = 0.01 * (1 + (i % 3));
                gS.boundaryField()[p][i] = symmTensor(g, a, -a, 1.2 * g, 0.5 * a, 0.9 * g);
                gT.boundaryField()[p][i] = tensor(g, a, -a, a, 1.2 * g, 0.5 * a, -0.5 * a, 0.25 * a, 0.9 * g);
            }
        put("gammaS", gS.internalField());
        put("gammaT", gT.internalField());
        forAll(mesh.boundary(), p)
        {
            put(pname(p, "gammaS").c_str(), gS.boundaryField()[p]);
            put(pname(p, "gammaT").c_str(), gT.boundaryField()[p]);
        }
        fv::gaussLaplacianScheme<scalar, symmTensor> lS
        (
            mesh, tmp<surfaceInterpolationScheme<symmTensor> >(new linear<symmTensor>(mesh)),
            tmp<fv::snGradScheme<scalar> >(new fv::correctedSnGrad<scalar>(mesh))
        );
        tmp<fvScalarMatrix> MS = lS.fvmLaplacian(gS, T);
        put("ref_lapS_upper", MS().upper());
        put("ref_lapS_diag", MS().diag());
        put("ref_lapS_source", MS().source());
        forAll(mesh.boundary(), p)
        {
            put(pname(p, "ref_lapS_internalCoeffs").c_str(), MS().internalCoeffs()[p]);
            put(pname(p, "ref_lapS_boundaryCoeffs").c_str(), MS().boundaryCoeffs()[p]);
        }
        put("ref_fvcLaplacianS", lS.fvcLaplacian(gS, T)().internalField());
        fv::gaussLaplacianScheme<scalar, tensor> lT
        (
            mesh, tmp<surfaceInterpolationScheme<tensor> >(new linear<tensor>(mesh)),
            tmp<fv::snGradScheme<scalar> >(new fv::correctedSnGrad<scalar>(mesh))
        );
        tmp<fvScalarMatrix> MT = lT.fvmLaplacian(gT, T);
        put("ref_lapT_upper", MT().upper());
        put("ref_lapT_diag", MT().diag());
        put("ref_lapT_source", MT().source());
    }
    // ---- a39: gaussDivScheme::fvcDiv (gaussDivScheme.C:48-68) of a vector and of a tensor field
    {
        fv::gaussDivScheme<vector> dv(mesh);
        put("ref_divU", dv.fvcDiv(U)().internalField());
        fv::gaussDivScheme<tensor> dt(mesh);
        put("ref_divGradU", dt.fvcDiv(gradU)().internalField());
    }
    fclose(out);
    return 0;
}

// Config C2: the pressure equation of simpleFoam's pEqn.H on the pitzDaily mesh the reference's own
// blockMesh produced - fvm::laplacian(rAU, p) == fvc::div(phiHbyA) with the tutorial's schemes (Gauss linear
// corrected, snGrad corrected: system/fvSchemes of the case) and boundary conditions (0/p: outlet fixedValue 0,
// everything else zeroGradient, frontAndBack empty), solved by the reference's own fvScalarMatrix::solve with the
// dictionary passed on the command line (the motorBike GAMG block).  rAU and HbyA are analytic fields of the cell
// centres (simpleFoam itself - turbulence libraries - is not built here).
static int pitz(fvMesh& mesh, Time& runTime, const char* solverDict)
{
    const label nC = mesh.nCells();
    dimensionSet::debug = 0;
    wordList types(mesh.boundary().size());
    forAll(types, p)
    {
        const fvPatch& fp = mesh.boundary()[p];
        types[p] = fp.type() == "empty" ? word("empty") : (fp.name() == "outlet" ? word("fixedValue") : word("zeroGradient"));
    }
    volScalarField p(IOobject("p", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0), types);
    wordList calcTypes(mesh.boundary().size());
    forAll(calcTypes, pi) calcTypes[pi] = mesh.boundary()[pi].type() == "empty" ? word("empty") : word("calculated");
    volScalarField rAU(IOobject("rAU", runTime.timeName(), mesh), mesh, dimensionedScalar("0", dimless, 0.0), calcTypes);
    volVectorField HbyA(IOobject("HbyA", runTime.timeName(), mesh), mesh, dimensionedVector("0", dimless, vector::zero), calcTypes);
    const vectorField& C = mesh.C().internalField();
    for (label c = 0; c < nC; c++)
    {
        const scalar x = 40.0 * C[c].x(), y = 40.0 * C[c].y();
        rAU.internalField()[c] = 1.0 + 0.3 * Foam::sin(1.7 * x) * Foam::cos(2.3 * y);
        HbyA.internalField()[c] = vector(10.0 * (1.0 - 0.6 * y * y), 0.8 * Foam::sin(2.0 * x + y), 0.0);
        // the pressure of the previous SIMPLE iteration: a non-trivial field, so that the explicit non-orthogonal
        // correction (correctedSnGrad of grad(p)) is at work and the solver starts from a non-zero guess
        p.internalField()[c] = 3.0 * Foam::cos(0.13 * x) - 0.4 * y + 0.02 * x * y;
    }
    p.correctBoundaryConditions();
    put("p0", p.internalField());
    forAll(mesh.boundary(), pi)
    {
        if (mesh.boundary()[pi].type() == "empty") continue;
        const vectorField& Cf = mesh.Cf().boundaryField()[pi];
        const bool wall = mesh.boundary()[pi].type() == "wall";
        forAll(Cf, i)
        {
            const scalar x = 40.0 * Cf[i].x(), y = 40.0 * Cf[i].y();
            rAU.boundaryField()[pi][i] = 1.0 + 0.3 * Foam::sin(1.7 * x) * Foam::cos(2.3 * y);
            HbyA.boundaryField()[pi][i] = wall ? vector::zero : vector(10.0 * (1.0 - 0.6 * y * y), 0.8 * Foam::sin(2.0 * x + y), 0.0);
        }
    }
    surfaceScalarField rAUf("rAUf", linear<scalar>(mesh).interpolate(rAU));
    surfaceScalarField phiHbyA("phiHbyA", linear<vector>(mesh).interpolate(HbyA) & mesh.Sf());
    fv::gaussLaplacianScheme<scalar, scalar> ls
    (
        mesh, tmp<surfaceInterpolationScheme<scalar> >(new linear<scalar>(mesh)),
        tmp<fv::snGradScheme<scalar> >(new fv::correctedSnGrad<scalar>(mesh))
    );
    tmp<fvScalarMatrix> tLap = ls.fvmLaplacian(rAUf, p);
    fvScalarMatrix pEqn(tLap() == fvc::surfaceIntegrate(phiHbyA));

    put("rAUf", rAUf.internalField());
    put("phiHbyA", phiHbyA.internalField());
    put("upper", pEqn.upper());
    put("diag", pEqn.diag());
    put("source", pEqn.source());
    put("V", mesh.V().field());
    put("Sf", mesh.Sf().internalField());
    put("magSf", mesh.magSf().internalField());
    put("C", C);
    put("weights", mesh.weights().internalField());
    put("nonOrthDeltaCoeffs", mesh.nonOrthDeltaCoeffs().internalField());
    put("nonOrthCorrectionVectors", mesh.nonOrthCorrectionVectors().internalField());
    {
        scalarField np(1, scalar(mesh.boundary().size()));
        put("nPatches", np);
    }
    forAll(mesh.boundary(), pi)
    {
        const labelUList& fc = mesh.boundary()[pi].faceCells();
        scalarField fcd(fc.size());
        forAll(fc, i) fcd[i] = fc[i];
        put(pname(pi, "faceCells").c_str(), fcd);
        put(pname(pi, "internalCoeffs").c_str(), pEqn.internalCoeffs()[pi]);
        put(pname(pi, "boundaryCoeffs").c_str(), pEqn.boundaryCoeffs()[pi]);
        put(pname(pi, "phiHbyA").c_str(), phiHbyA.boundaryField()[pi]);
        put(pname(pi, "rAUf").c_str(), rAUf.boundaryField()[pi]);
        put(pname(pi, "magSf").c_str(), mesh.magSf().boundaryField()[pi]);
        put(pname(pi, "deltaCoeffs").c_str(), mesh.deltaCoeffs().boundaryField()[pi]);
        put(pname(pi, "p0").c_str(), p.boundaryField()[pi]);
        put(pname(pi, "Sf").c_str(), mesh.Sf().boundaryField()[pi]);
    }
    if (pEqn.faceFluxCorrectionPtr()) put("faceFluxCorrection", pEqn.faceFluxCorrectionPtr()->internalField());
    {
        scalarField w(mag(cmptMultiply(mesh.Sf().internalField() / sqrt(mesh.magSf().internalField()),
                                       vector(1, 1.01, 1.02))));
        put("faceAreaPairWeights", w);
    }
    solverPerformance::debug = 2;   // per-iteration residual lines (SolverPerformance.C:65-71)
    IStringStream dictStream(solverDict);
    dictionary d(dictStream);
    solverPerformance perf = pEqn.solve(d);
    scalarField pf(5);
    pf[0] = perf.initialResidual(); pf[1] = perf.finalResidual(); pf[2] = perf.nIterations(); pf[3] = perf.converged();
    pf[4] = perf.singular();
    put("ref_perf", pf);
    put("ref_psi", p.internalField());
    // the flux the application takes from the solved equation (pEqn.H: phi = phiHbyA - pEqn.flux())
    {
        tmp<surfaceScalarField> fl = pEqn.flux();
        put("ref_flux", fl().internalField());
    }
    fclose(out);
    return 0;
}

int main(int argc, char* argv[])
{
    if (argc < 4 || argc > 6) { fprintf(stderr, "usage: fv_driver caseDir in.bin out.bin [stencils|glue|glueV|solve|solve2] [extra GAMG controls for solve]\n"); return 2; }
    fileName caseDir(argv[1]);
    Time runTime(Time::controlDictName, fileName(caseDir.path()), fileName(caseDir.name()));
    fvMesh mesh(IOobject(fvMesh::defaultRegion, runTime.timeName(), runTime, IOobject::MUST_READ));
    const label nC = mesh.nCells();
    const label nF = mesh.nInternalFaces();

    if (argc == 6 && std::string(argv[4]) == "pitz")
    {
        out = fopen(argv[3], "wb");
        return pitz(mesh, runTime, argv[5]);
    }
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
    if (argc >= 5 && std::string(argv[4]) == "glue") return glue(mesh, runTime, in);
    if (argc >= 5 && std::string(argv[4]) == "glueV") return glueV(mesh, runTime, in);
    if (argc >= 5 && std::string(argv[4]) == "nonorth") return nonorth(mesh, runTime, in);
    if (argc == 5 && std::string(argv[4]) == "solve") return solveMode(mesh, runTime, in, "nCellsInCoarsestLevel 10;");
    if (argc == 6 && std::string(argv[4]) == "solve") return solveMode(mesh, runTime, in, argv[5]);
    // two identical halves coupled by a cyclic pair = serial emulation of a 2-rank run: the combined
    // coarsest-level criterion 2n equals the and-reduced per-rank criterion n
    if (argc == 5 && std::string(argv[4]) == "solve2") return solveMode(mesh, runTime, in, "nCellsInCoarsestLevel 20;");

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
    // ---- 8f rank 2: linearUpwind correction and cellLimited Gauss linear gradient (scalar)
    {
        put("C", mesh.C().internalField());
        put("Cf", mesh.Cf().internalField());
        scalarField nb(1, scalar(mesh.boundary().size()));
        put("nPatches", nb);
        forAll(mesh.boundary(), p)
        {
            const labelUList& fc = mesh.boundary()[p].faceCells();
            scalarField fcd(fc.size());
            forAll(fc, i) fcd[i] = fc[i];
            char nm[64];
            snprintf(nm, sizeof(nm), "p%d_faceCells", p); put(nm, fcd);
            snprintf(nm, sizeof(nm), "p%d_Cf", p); put(nm, mesh.Cf().boundaryField()[p]);
            snprintf(nm, sizeof(nm), "p%d_value", p); put(nm, vf.boundaryField()[p]);
        }
        IStringStream lu("grad(vf)");
        linearUpwind<scalar> sch(mesh, phi, lu);
        tmp<surfaceScalarField> corr = sch.correction(vf);
        put("linearUpwind_correction", corr().internalField());
        {
            IStringStream gs("Gauss linear");
            tmp<volVectorField> g0 = fv::gradScheme<scalar>::New(mesh, gs)().calcGrad(vf, "g0");
            put("gaussLinearGrad", g0().internalField());
        }
        const char* ks[2] = {"1", "0.5"};
        for (int i = 0; i < 2; i++)
        {
            IStringStream cl((std::string("Gauss linear ") + ks[i]).c_str());
            fv::cellLimitedGrad<scalar> clg(mesh, cl);
            tmp<volVectorField> g = clg.calcGrad(vf, "g");
            put(i ? "cellLimitedGrad_k05" : "cellLimitedGrad_k1", g().internalField());
        }
    }
    // ---- 8f rank 2, vector forms (what motorBike's fvSchemes selects for U): linearUpwindV correction and
    //      cellLimited Gauss linear gradient of a vector field, non-zero boundary values
    {
        forAll(U.boundaryField(), p)
        {
            forAll(U.boundaryField()[p], i)
            {
                U.boundaryField()[p][i] = vector(0.3*((i % 4) - 1.5), 0.2*((i % 3) - 1.0), 0.1*((i + p) % 5));
            }
            char nm[64];
            snprintf(nm, sizeof(nm), "p%d_valueU", p); put(nm, U.boundaryField()[p]);
        }
        {
            IStringStream gs("Gauss linear");
            tmp<volTensorField> g0 = fv::gradScheme<vector>::New(mesh, gs)().calcGrad(U, "g0U");
            put("gaussLinearGradU", g0().internalField());
        }
        IStringStream lu("grad(U)");
        linearUpwindV<vector> sch(mesh, phi, lu);
        tmp<surfaceVectorField> corr = sch.correction(U);
        put("linearUpwindV_correction", corr().internalField());
        const char* ks[2] = {"1", "0.5"};
        for (int i = 0; i < 2; i++)
        {
            IStringStream cl((std::string("Gauss linear ") + ks[i]).c_str());
            fv::cellLimitedGrad<vector> clg(mesh, cl);
            tmp<volTensorField> g = clg.calcGrad(U, "gU");
            put(i ? "cellLimitedGradV_k05" : "cellLimitedGradV_k1", g().internalField());
        }
    }
    // ---- the `bounded` wrapper of motorBike's div schemes (boundedConvectionScheme.C:60-77):
    //      scheme.fvmDiv(phi, vf) - fvm::Sp(fvc::surfaceIntegrate(phi), vf), with non-zero boundary fluxes
    {
        dimensionSet::debug = 0;
        forAll(phi.boundaryField(), p)
        {
            forAll(phi.boundaryField()[p], i) phi.boundaryField()[p][i] = 0.04*((i % 5) - 2) + 0.01*p;
            char nm[64];
            snprintf(nm, sizeof(nm), "p%d_phi", p); put(nm, phi.boundaryField()[p]);
        }
        fv::gaussConvectionScheme<scalar> cu(mesh, phi,
            tmp<surfaceInterpolationScheme<scalar> >(new upwind<scalar>(mesh, phi)));
        tmp<fvScalarMatrix> Mu = cu.fvmDiv(phi, vfz);
        put("div_upwind_diag_bphi", Mu().diag());
        fvScalarMatrix Mb(Mu() - fvm::Sp(fvc::surfaceIntegrate(phi), vfz));
        put("div_bounded_upwind_diag", Mb.diag());
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
