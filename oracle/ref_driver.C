/*
 * TEST INFRASTRUCTURE (oracle/_ref) - driver for the REFERENCE's own libOpenFOAM.
 *
 * Our code; compiled against the reference headers where they lie under
 * /root/reference and linked with oracle/_ref/libOpenFOAM.so (see build_ref.sh).
 * It builds an lduPrimitiveMesh + lduMatrix from plain arrays and calls the
 * reference's own run-time-selected solver / preconditioner / smoother /
 * agglomeration (src/OpenFOAM/matrices/lduMatrix/...), then dumps the results
 * so that (a) the C restatement in oracle/ldu_oracle.c can be pinned against
 * the real thing and (b) golden vectors can be generated (tests/golden/).
 *
 * usage:  ref_driver <mode> <problem.ldub> <out.ldub> <caseDir> ["dict string"]
 *   mode = solve   : lduMatrix::solver::New(...)->solve(psi, source)
 *   mode = time    : the same, twice, timed, silent (bench.py cpu_baseline kind "reference")
 *          ops     : Amul/Tmul/sumA/residual/preconditioners/smoothers on psi, source
 *          agglom  : GAMG agglomeration + level matrices dump
 *          rcm     : Foam::bandCompression on the cell-cell addressing of the faces (what renumberMesh applies)
 *          cops    : LduMatrix<vector,scalar,scalar> Amul/Tmul/residual/preconditioners/smoother on psiV, sourceV
 *          csolve  : LduMatrix<vector,scalar,scalar>::solver::New(...)->solve(psiV)   ("type coupled")
 *
 * Container format ("LDUB"): records {char name[32]; int32 dtype(0=i32,1=f64);
 * int64 count; payload}.
 */

#include "lduPrimitiveMesh.H"
#include "lduMatrix.H"
#include "Time.H"
#include "IStringStream.H"
#include "GAMGSolver.H"
#include "GAMGAgglomeration.H"
#include "pairGAMGAgglomeration.H"
#include "addToRunTimeSelectionTable.H"
#include "DICPreconditioner.H"
#include "LduMatrix.H"
#include "bandCompression.H"
#include "vector.H"
#include "vectorField.H"
#include "symmTensorField.H"
#include "OSspecific.H"

#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <time.h>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <stdint.h>

using namespace Foam;

// ---------------------------------------------------------------- container

struct Rec
{
    int dtype;
    std::vector<int> i;
    std::vector<double> d;
};
typedef std::map<std::string, Rec> Recs;

static Recs readLdub(const char* fn)
{
    Recs r;
    FILE* f = fopen(fn, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", fn); exit(2); }
    char name[32];
    while (fread(name, 1, 32, f) == 32)
    {
        int32_t dt; int64_t n;
        if (fread(&dt, 4, 1, f) != 1 || fread(&n, 8, 1, f) != 1) break;
        Rec rec; rec.dtype = dt;
        if (dt == 0) { rec.i.resize(n); if (n) (void)!fread(&rec.i[0], 4, n, f); }
        else         { rec.d.resize(n); if (n) (void)!fread(&rec.d[0], 8, n, f); }
        name[31] = 0;
        r[name] = rec;
    }
    fclose(f);
    return r;
}

static FILE* outF = 0;

static void putI(const char* nm, const int* p, int64_t n)
{
    char name[32]; memset(name, 0, 32); strncpy(name, nm, 31);
    int32_t dt = 0;
    fwrite(name, 1, 32, outF); fwrite(&dt, 4, 1, outF); fwrite(&n, 8, 1, outF);
    if (n) fwrite(p, 4, n, outF);
}
static void putD(const char* nm, const double* p, int64_t n)
{
    char name[32]; memset(name, 0, 32); strncpy(name, nm, 31);
    int32_t dt = 1;
    fwrite(name, 1, 32, outF); fwrite(&dt, 4, 1, outF); fwrite(&n, 8, 1, outF);
    if (n) fwrite(p, 8, n, outF);
}
static void putL(const char* nm, const labelUList& l) { putI(nm, l.begin(), l.size()); }
static void putS(const char* nm, const scalarField& s) { putD(nm, s.begin(), s.size()); }

// ---------------------------------------------------------------- mesh with a registry

class regLduMesh
:
    public lduPrimitiveMesh
{
    const objectRegistry& db_;
public:
    regLduMesh
    (
        const objectRegistry& db, label nCells, const labelUList& l, const labelUList& u,
        const labelListList& pa, lduInterfacePtrsList ifs, const lduSchedule& ps
    )
    :
        lduPrimitiveMesh(nCells, l, u, pa, ifs, ps), db_(db)
    {}
    virtual const objectRegistry& thisDb() const { return db_; }
};

// ---------------------------------------------------------------- agglomerator with supplied weights
// Runs the reference's pairGAMGAgglomeration::agglomerate(mesh, weights) with
// face weights supplied by the problem file (what faceAreaPair computes from
// Sf in libfiniteVolume:  |Sf/sqrt(magSf) * (1,1.01,1.02)|).

static scalarField* suppliedWeights = 0;

namespace Foam
{
class suppliedPairGAMGAgglomeration
:
    public pairGAMGAgglomeration
{
public:
    TypeName("faceAreaPair");
    suppliedPairGAMGAgglomeration(const lduMatrix& m, const dictionary& d)
    :
        pairGAMGAgglomeration(m.mesh(), d)
    {
        if (!suppliedWeights)
        {
            FatalErrorIn("suppliedPairGAMGAgglomeration") << "no faceWeights in problem"
                << exit(FatalError);
        }
        agglomerate(m.mesh(), *suppliedWeights);
    }
};
defineTypeNameAndDebug(suppliedPairGAMGAgglomeration, 0);
addToRunTimeSelectionTable(GAMGAgglomeration, suppliedPairGAMGAgglomeration, lduMatrix);
}

static dictionary mkDict(const char* s)
{
    IStringStream is(s);
    return dictionary(is);
}

// ---------------------------------------------------------------- main

int main(int argc, char* argv[])
{
    if (argc < 5)
    {
        fprintf(stderr, "usage: ref_driver mode problem out caseDir [dict]\n");
        return 2;
    }
    const std::string mode = argv[1];
    Recs P = readLdub(argv[2]);
    outF = fopen(argv[3], "wb");
    fileName caseDir(argv[4]);
    const char* dictStr = argc > 5 ? argv[5] : "";

    // full-precision residual prints
    IOstream::defaultPrecision(17);
    Sout.precision(17);
    Serr.precision(17);

    // minimal case for the Time registry
    mkDir(caseDir/"system");
    mkDir(caseDir/"constant");
    {
        FILE* f = fopen((caseDir/"system"/"controlDict").c_str(), "w");
        fprintf(f,
            "FoamFile { version 2.0; format ascii; class dictionary; object controlDict; }\n"
            "application none; startFrom startTime; startTime 0; stopAt endTime; endTime 1;\n"
            "deltaT 1; writeControl timeStep; writeInterval 1000000; writeFormat ascii;\n"
            "writePrecision 17; writeCompression off; timeFormat general; timePrecision 6;\n"
            "runTimeModifiable false;\n");
        // optional plugin (the product's OpenFOAM shim): loaded by the reference's own
        // dlLibraryTable exactly like `libs (...)` in a user's system/controlDict (Time.C:343)
        if (getenv("LDU_PLUGIN_LIB")) fprintf(f, "libs (\"%s\");\n", getenv("LDU_PLUGIN_LIB"));
        fclose(f);
    }
    Time runTime(Time::controlDictName, caseDir.path(), caseDir.name(), "system", "constant", false);

    IOstream::defaultPrecision(17);
    Sout.precision(17);

    const label nCells = P["nCells"].i[0];
    const label nFaces = P["lowerAddr"].i.size();
    labelList l(nFaces), u(nFaces);
    for (label f = 0; f < nFaces; f++) { l[f] = P["lowerAddr"].i[f]; u[f] = P["upperAddr"].i[f]; }

    lduSchedule sched(0);
    regLduMesh mesh(runTime, nCells, l, u, labelListList(0), lduInterfacePtrsList(0), sched);

    lduMatrix A(mesh);
    {
        scalarField& d = A.diag();
        for (label c = 0; c < nCells; c++) d[c] = P["diag"].d[c];
        scalarField& up = A.upper();
        for (label f = 0; f < nFaces; f++) up[f] = P["upper"].d[f];
        if (P.count("lower"))
        {
            scalarField& lo = A.lower();
            for (label f = 0; f < nFaces; f++) lo[f] = P["lower"].d[f];
        }
    }
    scalarField psi(nCells, 0.0), source(nCells, 0.0);
    if (P.count("psi"))    for (label c = 0; c < nCells; c++) psi[c] = P["psi"].d[c];
    if (P.count("source")) for (label c = 0; c < nCells; c++) source[c] = P["source"].d[c];
    scalarField weights;
    if (P.count("faceWeights"))
    {
        weights.setSize(nFaces);
        for (label f = 0; f < nFaces; f++) weights[f] = P["faceWeights"].d[f];
        suppliedWeights = &weights;
    }

    FieldField<Field, scalar> bc(0), ic(0);
    lduInterfaceFieldPtrsList ifs(0);

    if (mode == "solve")
    {
        dictionary dict(mkDict(dictStr));
        lduMatrix::debug = 2;
        solverPerformance::debug = 2;
        GAMGSolver::debug = 2;
        solverPerformance perf =
            lduMatrix::solver::New("p", A, bc, ic, ifs, dict)->solve(psi, source);
        perf.print(Info);
        putS("psi", psi);
        double pv[5] =
        {
            perf.initialResidual(), perf.finalResidual(), double(perf.nIterations()),
            double(perf.converged()), double(perf.singular())
        };
        putD("perf", pv, 5);
    }
    else if (mode == "time")
    {
        // CPU baseline of bench.py: the reference's own solver timed on this host, no debug output.
        // Two solves from the same initial guess: the first builds (and, with cacheAgglomeration on,
        // caches) the agglomeration, the second is the steady-state cost bench.py compares against.
        dictionary dict(mkDict(dictStr));
        const scalarField psi0(psi);
        double secs[2] = {0, 0};
        double its[2] = {0, 0};
        for (int rep = 0; rep < 2; rep++)
        {
            psi = psi0;
            timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            solverPerformance perf =
                lduMatrix::solver::New("p", A, bc, ic, ifs, dict)->solve(psi, source);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            secs[rep] = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
            its[rep] = perf.nIterations();
        }
        double tv[4] = {secs[0], secs[1], its[0], its[1]};
        putD("time", tv, 4);
        putS("psi", psi);
    }
    else if (mode == "ops")
    {
        scalarField y(nCells);
        A.Amul(y, psi, bc, ifs, 0);            putS("Amul", y);
        A.Tmul(y, psi, ic, ifs, 0);            putS("Tmul", y);
        A.sumA(y, bc, ifs);                    putS("sumA", y);
        A.residual(y, psi, source, bc, ifs, 0); putS("residual", y);
        {
            tmp<scalarField> tH1 = A.H1();     putS("H1", tH1());
            tmp<Field<scalar> > tH = A.H(psi); putS("H", tH());
            tmp<Field<scalar> > tf = A.faceH(psi); putS("faceH", tf());
        }
        putL("losort", mesh.lduAddr().losortAddr());
        putL("ownerStart", mesh.lduAddr().ownerStartAddr());
        putL("losortStart", mesh.lduAddr().losortStartAddr());

        const char* symPre[]  = {"DIC", "FDIC", "diagonal", "none", 0};
        const char* asymPre[] = {"DILU", "diagonal", "none", 0};
        const char** pre = A.symmetric() ? symPre : asymPre;
        const char* solName = A.symmetric() ? "PCG" : "PBiCG";
        for (; *pre; ++pre)
        {
            std::string ds = std::string("solver ") + solName + "; preconditioner " + *pre + ";";
            dictionary dict(mkDict(ds.c_str()));
            autoPtr<lduMatrix::solver> sol = lduMatrix::solver::New("p", A, bc, ic, ifs, dict);
            autoPtr<lduMatrix::preconditioner> pc = lduMatrix::preconditioner::New(sol(), dict);
            scalarField w(nCells, 0.0);
            pc->precondition(w, source, 0);
            putS((std::string("precond_") + *pre).c_str(), w);
            if (!A.symmetric() && std::string(*pre) != "none")
            {
                scalarField wT(nCells, 0.0);
                pc->preconditionT(wT, source, 0);
                putS((std::string("precondT_") + *pre).c_str(), wT);
            }
        }
        if (A.symmetric())
        {
            scalarField rD(A.diag());
            DICPreconditioner::calcReciprocalD(rD, A);
            putS("rD_DIC", rD);
        }
        const char* symSm[]  = {"GaussSeidel", "symGaussSeidel", "DIC", "FDIC", "DICGaussSeidel", 0};
        const char* asymSm[] = {"GaussSeidel", "symGaussSeidel", "DILU", "DILUGaussSeidel", 0};
        const char** sm = A.symmetric() ? symSm : asymSm;
        for (; *sm; ++sm)
        {
            std::string ds = std::string("smoother ") + *sm + ";";
            dictionary dict(mkDict(ds.c_str()));
            autoPtr<lduMatrix::smoother> s = lduMatrix::smoother::New("p", A, bc, ic, ifs, dict);
            scalarField x(psi);
            s->smooth(x, source, 0, 1);
            putS((std::string("smooth1_") + *sm).c_str(), x);
            s->smooth(x, source, 0, 2);
            putS((std::string("smooth3_") + *sm).c_str(), x);
        }
    }
    else if (mode == "agglom")
    {
        dictionary dict(mkDict(dictStr));
        GAMGSolver gs("p", A, bc, ic, ifs, dict);
        const GAMGAgglomeration& ag = gs.agglomeration_;
        int nLevels = ag.size();
        putI("nLevels", &nLevels, 1);
        for (label lev = 0; lev < nLevels; lev++)
        {
            char nm[32];
            sprintf(nm, "restrict_%d", lev);     putL(nm, ag.restrictAddressing(lev));
            sprintf(nm, "faceRestrict_%d", lev); putL(nm, ag.faceRestrictAddressing(lev));
            const lduAddressing& ca = ag.meshLevel(lev + 1).lduAddr();
            int nc = ca.size();
            sprintf(nm, "nCells_%d", lev);       putI(nm, &nc, 1);
            sprintf(nm, "lowerAddr_%d", lev);    putL(nm, ca.lowerAddr());
            sprintf(nm, "upperAddr_%d", lev);    putL(nm, ca.upperAddr());
            const lduMatrix& cm = gs.matrixLevels_[lev];
            sprintf(nm, "diag_%d", lev);         putS(nm, cm.diag());
            sprintf(nm, "upper_%d", lev);        putS(nm, cm.upper());
            if (cm.hasLower())
            {
                sprintf(nm, "lower_%d", lev);    putS(nm, cm.lower());
            }
        }
    }
    else if (mode == "rcm")
    {
        // primitiveMesh::calcCellCells (primitiveMeshCellCells.C): face order, own learns nei, nei learns own
        labelList n(nCells, 0);
        for (label f = 0; f < nFaces; f++) { n[l[f]]++; n[u[f]]++; }
        labelListList cc(nCells);
        for (label c = 0; c < nCells; c++) cc[c].setSize(n[c]);
        n = 0;
        for (label f = 0; f < nFaces; f++)
        {
            cc[l[f]][n[l[f]]++] = u[f];
            cc[u[f]][n[u[f]]++] = l[f];
        }
        labelList order(bandCompression(cc));
        putL("newOrder", order);
    }
    else if (mode == "csolve6")
    {
        // the same family on a symmTensor field (six components): LduMatrix<symmTensor, scalar, scalar>
        typedef LduMatrix<symmTensor, scalar, scalar> sMatrix;
        sMatrix M(mesh);
        M.diag() = A.diag();
        M.upper() = A.upper();
        if (A.hasLower()) M.lower() = A.lower();
        symmTensorField psiS(nCells, symmTensor::zero), sourceS(nCells, symmTensor::zero);
        for (label c = 0; c < nCells; c++)
            for (direction k = 0; k < 6; k++)
            {
                if (P.count("psiV")) psiS[c][k] = P["psiV"].d[6*c + k];
                if (P.count("sourceV")) sourceS[c][k] = P["sourceV"].d[6*c + k];
            }
        M.source() = sourceS;
        dictionary dict(mkDict(dictStr));
        SolverPerformance<symmTensor> perf = sMatrix::solver::New("R", M, dict)->solve(psiS);
        putD("psiV", reinterpret_cast<const double*>(psiS.begin()), 6*nCells);
        double pv[15];
        for (direction k = 0; k < 6; k++)
        {
            pv[k] = perf.initialResidual()[k];
            pv[6 + k] = perf.finalResidual()[k];
        }
        pv[12] = perf.nIterations();
        pv[13] = perf.converged();
        pv[14] = perf.singular();
        putD("perf", pv, 15);
    }
    else if (mode == "cops" || mode == "csolve")
    {
        // the templated coupled family on the same coefficients (fvMatrixSolve.C:222-277 builds exactly this)
        typedef LduMatrix<vector, scalar, scalar> cMatrix;
        cMatrix M(mesh);
        M.diag() = A.diag();
        M.upper() = A.upper();
        if (A.hasLower()) M.lower() = A.lower();
        vectorField psiV(nCells, vector::zero), sourceV(nCells, vector::zero);
        for (label c = 0; c < nCells; c++)
            for (direction k = 0; k < 3; k++)
            {
                if (P.count("psiV")) psiV[c][k] = P["psiV"].d[3*c + k];
                if (P.count("sourceV")) sourceV[c][k] = P["sourceV"].d[3*c + k];
            }
        M.source() = sourceV;
        if (mode == "csolve")
        {
            dictionary dict(mkDict(dictStr));
            cMatrix::debug = 2;
            SolverPerformance<vector>::debug = 2;
            SolverPerformance<vector> perf = cMatrix::solver::New("U", M, dict)->solve(psiV);
            perf.print(Info);
            putD("psiV", reinterpret_cast<const double*>(psiV.begin()), 3*nCells);
            double pv[11];
            for (direction k = 0; k < 3; k++)
            {
                pv[k] = perf.initialResidual()[k];
                pv[3 + k] = perf.finalResidual()[k];
            }
            pv[6] = perf.nIterations();
            pv[7] = perf.converged();
            pv[8] = perf.singular();
            putD("perf", pv, 9);
        }
        else
        {
            vectorField y(nCells, vector::zero);
            M.Amul(y, psiV);     putD("Amul", reinterpret_cast<const double*>(y.begin()), 3*nCells);
            M.Tmul(y, psiV);     putD("Tmul", reinterpret_cast<const double*>(y.begin()), 3*nCells);
            M.residual(y, psiV); putD("residual", reinterpret_cast<const double*>(y.begin()), 3*nCells);
            const char* symPre[]  = {"diagonal", "none", 0};
            const char* asymPre[] = {"DILU", "diagonal", "none", 0};
            const char** pre = M.symmetric() ? symPre : asymPre;
            const char* solName = M.symmetric() ? "PCICG" : "PBiCICG";
            for (; *pre; ++pre)
            {
                std::string ds = std::string("solver ") + solName + "; preconditioner " + *pre + ";";
                dictionary dict(mkDict(ds.c_str()));
                autoPtr<cMatrix::solver> sol = cMatrix::solver::New("U", M, dict);
                autoPtr<cMatrix::preconditioner> pc = cMatrix::preconditioner::New(sol(), dict);
                vectorField w(nCells, vector::zero);
                pc->precondition(w, sourceV);
                putD((std::string("precond_") + *pre).c_str(), reinterpret_cast<const double*>(w.begin()), 3*nCells);
                if (!M.symmetric())
                {
                    vectorField wT(nCells, vector::zero);
                    pc->preconditionT(wT, sourceV);
                    putD((std::string("precondT_") + *pre).c_str(), reinterpret_cast<const double*>(wT.begin()),
                         3*nCells);
                }
            }
            {
                dictionary dict(mkDict("smoother GaussSeidel;"));
                autoPtr<cMatrix::smoother> sm = cMatrix::smoother::New("U", M, dict);
                vectorField x(psiV);
                sm->smooth(x, 1);
                putD("smooth1_GaussSeidel", reinterpret_cast<const double*>(x.begin()), 3*nCells);
                sm->smooth(x, 2);
                putD("smooth3_GaussSeidel", reinterpret_cast<const double*>(x.begin()), 3*nCells);
            }
        }
    }
    else
    {
        fprintf(stderr, "unknown mode %s\n", mode.c_str());
        return 2;
    }
    fclose(outF);
    return 0;
}
