// TEST / BENCH-INPUT INFRASTRUCTURE (never shipped, never linked by the product): a lduMatrix::solver for the reference's own
// run-time selection table, "dumpGAMG", that writes the matrix it is handed - in the format of the shim's LDU_DUMP_MATRIX
// (openfoam-2.2.x_amd/plugin/hipLduSolvers.C: hipDumpMatrix; read back by tests/test_simplefoam_motorbike.py: read_dump) - and
// then lets the REFERENCE's GAMGSolver (GAMGSolver.C:44-127, GAMGSolverSolve.C:34-117) solve it, unchanged.  It exists so
// that the p-matrix of a real SIMPLE iteration on a mesh of millions of cells can be taken from the reference's simpleFoam
// HERE, on the CPU (the shim's dump needs the GPU).  Loaded through controlDict `libs ("libdumpSolver.so")`; selected with
// `solver dumpGAMG;` in fvSolution; LDU_DUMP_MATRIX="<field>:<n>:<file>" as for the shim.
// Build: oracle/build_dump_solver.sh (g++ against the headers oracle/build_ref.sh collected; output oracle/_ref/libdumpSolver.so).
#include "GAMGSolver.H"
#include "polyMesh.H"
#include "addToRunTimeSelectionTable.H"

#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

namespace Foam
{

class dumpGAMGSolver
:
    public lduMatrix::solver
{
public:

    TypeName("dumpGAMG");

    dumpGAMGSolver
    (
        const word& fieldName,
        const lduMatrix& matrix,
        const FieldField<Field, scalar>& interfaceBouCoeffs,
        const FieldField<Field, scalar>& interfaceIntCoeffs,
        const lduInterfaceFieldPtrsList& interfaces,
        const dictionary& solverControls
    )
    :
        lduMatrix::solver(fieldName, matrix, interfaceBouCoeffs, interfaceIntCoeffs, interfaces, solverControls)
    {}

    virtual ~dumpGAMGSolver() {}

    void dump(const scalarField& psi, const scalarField& source) const
    {
        static const char* spec = getenv("LDU_DUMP_MATRIX");
        if (!spec) return;
        static std::map<std::string, int> calls;
        const std::string s(spec);
        const size_t a = s.find(':'), b = s.find(':', a == std::string::npos ? a : a + 1);
        if (a == std::string::npos || b == std::string::npos) return;
        if (s.substr(0, a) != fieldName_) return;
        if (++calls[fieldName_] != atoi(s.substr(a + 1, b - a - 1).c_str())) return;
        FILE* f = fopen(s.substr(b + 1).c_str(), "wb");
        if (!f) return;
        const lduAddressing& ad = matrix_.lduAddr();
        const polyMesh* pm = dynamic_cast<const polyMesh*>(&matrix_.mesh());
        if (!pm) pm = dynamic_cast<const polyMesh*>(&matrix_.mesh().thisDb());
        const label nF = ad.lowerAddr().size();
        const bool sf = pm && pm->nInternalFaces() == nF;
        long long head[4] = {psi.size(), nF, matrix_.symmetric() || matrix_.diagonal(), sf};
        fwrite(head, sizeof(long long), 4, f);
        std::vector<int> idx(nF);
        forAll(ad.lowerAddr(), i) idx[i] = ad.lowerAddr()[i];
        fwrite(idx.data(), sizeof(int), nF, f);
        forAll(ad.upperAddr(), i) idx[i] = ad.upperAddr()[i];
        fwrite(idx.data(), sizeof(int), nF, f);
        fwrite(matrix_.diag().begin(), sizeof(double), psi.size(), f);
        fwrite(matrix_.upper().begin(), sizeof(double), nF, f);
        if (!head[2]) fwrite(matrix_.lower().begin(), sizeof(double), nF, f);
        fwrite(source.begin(), sizeof(double), psi.size(), f);
        fwrite(psi.begin(), sizeof(double), psi.size(), f);
        if (sf) fwrite(pm->faceAreas().begin(), sizeof(double), 3*size_t(nF), f);
        fclose(f);
        Info<< "[dumpGAMG] matrix of " << fieldName_ << " (call " << calls[fieldName_] << ") written to "
            << s.substr(b + 1).c_str() << endl;
    }

    virtual solverPerformance solve(scalarField& psi, const scalarField& source, const direction cmpt = 0) const
    {
        dump(psi, source);
        GAMGSolver gamg(fieldName_, matrix_, interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, controlDict_);
        return gamg.solve(psi, source, cmpt);
    }
};

defineTypeNameAndDebug(dumpGAMGSolver, 0);

lduMatrix::solver::addsymMatrixConstructorToTable<dumpGAMGSolver> adddumpGAMGSolverSymMatrixConstructorToTable_;

}
