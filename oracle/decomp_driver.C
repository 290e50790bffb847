// BENCH-INPUT INFRASTRUCTURE - runs the REFERENCE's own geometric decomposition methods (libdecompositionMethods, compiled
// from /root/reference by oracle/build_ref_mesh.sh: hierarchGeomDecomp / simpleGeomDecomp, what decomposePar calls for
// `method hierarchical | simple`, decomposePar/domainDecompositionDistribute.C:31-118) on a list of cell centres:
//   decomp_driver <centres.bin: N x 3 doubles> <N> <out.bin: N int32> "<decomposeParDict text>"
// e.g. the motorBike tutorial's system/decomposeParDict:17-33 (numberOfSubdomains 6; method hierarchical; n (3 2 1); delta
// 0.001; order xyz).  Our code; only reference HEADERS are included.  Never shipped.
#include "decompositionMethod.H"
#include "IStringStream.H"
#include "dictionary.H"
#include "pointField.H"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace Foam;

int main(int argc, char* argv[])
{
    if (argc != 5) { fprintf(stderr, "usage: decomp_driver centres.bin N out.bin \"dict\"\n"); return 2; }
    const long n = atol(argv[2]);
    pointField cc(n);
    {
        FILE* f = fopen(argv[1], "rb");
        if (!f) { perror(argv[1]); return 2; }
        std::vector<double> buf(3 * (size_t)n);
        if (fread(buf.data(), sizeof(double), buf.size(), f) != buf.size()) { fprintf(stderr, "short read\n"); return 2; }
        fclose(f);
        for (long i = 0; i < n; i++) cc[i] = point(buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]);
    }
    IStringStream is(argv[4]);
    dictionary dict(is);
    autoPtr<decompositionMethod> method(decompositionMethod::New(dict));
    labelList proc(method().decompose(cc));
    std::vector<int> out(n);
    labelList count(method().nDomains(), 0);
    for (long i = 0; i < n; i++) { out[i] = proc[i]; count[proc[i]]++; }
    FILE* g = fopen(argv[3], "wb");
    fwrite(out.data(), sizeof(int), out.size(), g);
    fclose(g);
    Info<< "decomp_driver: " << word(dict.lookup("method")) << " into " << method().nDomains() << " domains, cells per domain "
        << count << endl;
    return 0;
}
