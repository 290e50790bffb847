#include "greedy_reference.hpp"
#include "../../openfoam-2.2.x_amd/csrc/ldu_cluster_greedy.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char** argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 64;
    int irregular = argc > 2 ? atoi(argv[2]) : 0;
    long nC = (long)n * n * n;
    std::vector<int> l, u;
    auto id = [&](int i, int j, int k) { return (k * n + j) * n + i; };
    std::mt19937 rng(5);
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++)
    {
        int c = id(i, j, k);
        std::vector<int> nb;
        if (i + 1 < n) nb.push_back(id(i + 1, j, k));
        if (j + 1 < n) nb.push_back(id(i, j + 1, k));
        if (k + 1 < n) nb.push_back(id(i, j, k + 1));
        if (irregular && c + 1 < nC) { int extra = c + 1 + rng() % std::min<long>(nC - c - 1, 3000); bool dup = false; for (int x : nb) dup |= x == extra; if (!dup && rng() % 3 == 0) nb.push_back(extra); }
        std::sort(nb.begin(), nb.end());
        for (int x : nb) { l.push_back(c); u.push_back(x); }
    }
    int nF = (int)l.size();
    std::vector<int> ownerStart(nC + 1, 0), losortStart(nC + 1, 0), losort(nF), level(nC, 0);
    for (int f = 0; f < nF; f++) { ownerStart[l[f] + 1]++; losortStart[u[f] + 1]++; }
    for (long c = 0; c < nC; c++) { ownerStart[c + 1] += ownerStart[c]; losortStart[c + 1] += losortStart[c]; }
    { std::vector<int> pos(losortStart.begin(), losortStart.end() - 1); for (int f = 0; f < nF; f++) losort[pos[u[f]]++] = f; }
    for (int f = 0; f < nF; f++) level[u[f]] = std::max(level[u[f]], level[l[f]] + 1);   // faces are owner-sorted
    ClGreedyOld A; ClGreedy B;
    auto t0 = std::chrono::steady_clock::now();
    cluster_greedy_old((int)nC, nF, l.data(), u.data(), losort.data(), losortStart.data(), ownerStart.data(), level.data(), 64, A);
    auto t1 = std::chrono::steady_clock::now();
    cluster_greedy((int)nC, nF, l.data(), u.data(), losort.data(), losortStart.data(), ownerStart.data(), level.data(), 64, B);
    auto t2 = std::chrono::steady_clock::now();
    bool same = A.cluster == B.cluster && A.intra == B.intra && A.cLevel == B.cLevel && A.cDepth == B.cDepth && A.members.size() == B.nClusters();
    for (size_t i = 0; same && i < A.members.size(); i++)
    {
        same = (int)A.members[i].size() == B.memberStart[i + 1] - B.memberStart[i];
        for (size_t q = 0; same && q < A.members[i].size(); q++) same = A.members[i][q] == B.memberCells[B.memberStart[i] + q];
    }
    printf("n=%d irregular=%d cells %ld faces %d clusters %zu: old %.3f s new %.3f s identical %d\n", n, irregular, nC, nF, A.members.size(),
           std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count(), (int)same);
    return same ? 0 : 1;
}
