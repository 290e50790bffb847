// The greedy clustering as it was until round 4 (candidates in one vector, linear scan per pick): kept as the
// reference tools/greedy_check/harness.cpp compares the current ldu_cluster_greedy.hpp against, cluster for cluster.
// Greedy topological clustering of the lower-triangular dependency DAG (host only, no device calls): the plan
// step of the cluster sweep engine (ldu_cluster.hip), kept apart so that it can also run without a GPU
// (ldu_debug_dag_stats: plan statistics of an addressing on the build host).
#pragma once
#include <algorithm>
#include <queue>
#include <vector>

struct ClGreedyOld {
    std::vector<int> cluster;                 // [nC] cluster id in creation order
    std::vector<int> intra;                   // [nC] internal dependency level of the cell inside its cluster
    std::vector<std::vector<int>> members;    // cells of a cluster in the order they were absorbed (topological)
    std::vector<int> cLevel, cDepth;          // per cluster: level in the quotient DAG, internal steps
};

// Clusters are grown IN a topological order of the cell DAG: a cluster only absorbs "ready" cells (all lower
// neighbours placed), preferring the one with most neighbours already inside; seeds in (dependency level, index)
// order.  Any such partition has an acyclic quotient graph.
inline void cluster_greedy_old(int nC, int nF, const int* l, const int* u, const int* losort, const int* losortStart,
                           const int* ownerStart, const int* level, int maxCells, ClGreedyOld& G)
{
    std::vector<int> indeg(nC, 0);
    G.cluster.assign(nC, -1);
    G.intra.assign(nC, 0);
    G.members.clear(); G.cLevel.clear(); G.cDepth.clear();
    for (int f = 0; f < nF; f++) indeg[u[f]]++;
    // seeds in (dependency level, index) order: the clusters are created along the wavefront, so the
    // fragments left over where blobs do not tile (mesh dimensions that are no multiple of the blob size)
    // depend on their neighbours in parallel instead of forming one serial chain (54^3 box: 51 cluster
    // levels instead of 95 with index-ordered seeds; 40 would be ideal)
    typedef std::pair<int, int> Seed;
    std::priority_queue<Seed, std::vector<Seed>, std::greater<Seed>> ready;
    for (int c = 0; c < nC; c++) if (!indeg[c]) ready.push(Seed(level[c], c));
    std::vector<int> cand;
    // lower neighbours of a cell already inside the cluster being grown, kept incrementally (valid while
    // cntId[c] == id): the candidate scores without rescanning every candidate's neighbours at every pick
    std::vector<int> cnt(nC, 0), cntId(nC, -1);
    while (!ready.empty())
    {
        const int seed = ready.top().second; ready.pop();
        if (G.cluster[seed] >= 0) continue;
        const int id = (int)G.members.size();
        G.members.emplace_back();
        cand.clear(); cand.push_back(seed);
        int lev = 0, depth = 0;
        while (!cand.empty() && (int)G.members[id].size() < maxCells)
        {
            // most neighbours already inside; the earliest candidate wins ties
            int bi = 0, bscore = -1;
            for (size_t t = 0; t < cand.size(); t++)
            {
                const int c = cand[t];
                const int sc = cntId[c] == id ? cnt[c] : 0;
                if (sc > bscore) { bscore = sc; bi = (int)t; }
            }
            const int c = cand[bi];
            cand.erase(cand.begin() + bi);
            G.cluster[c] = id;
            G.members[id].push_back(c);
            int il = 0;
            for (int j = losortStart[c]; j < losortStart[c + 1]; j++)
            {
                const int p = l[losort[j]];
                if (G.cluster[p] == id) il = std::max(il, G.intra[p] + 1);
                else lev = std::max(lev, G.cLevel[G.cluster[p]] + 1);
            }
            G.intra[c] = il;
            depth = std::max(depth, il + 1);
            for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
            {
                const int v = u[f];
                if (cntId[v] != id) { cntId[v] = id; cnt[v] = 0; }
                cnt[v]++;
                if (--indeg[v] == 0) cand.push_back(v);
            }
        }
        for (int c : cand) ready.push(Seed(level[c], c));
        G.cLevel.push_back(lev);
        G.cDepth.push_back(depth);
    }
}
