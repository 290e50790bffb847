// Greedy topological clustering of the lower-triangular dependency DAG (host only, no device calls): the plan
// step of the cluster sweep engine (ldu_cluster.hip), kept apart so that it can also run without a GPU
// (ldu_debug_dag_stats: plan statistics of an addressing on the build host).
#pragma once
#include <algorithm>
#include <queue>
#include <vector>

struct ClGreedy {
    std::vector<int> cluster;                 // [nC] cluster id in creation order
    std::vector<int> intra;                   // [nC] internal dependency level of the cell inside its cluster
    std::vector<int> memberStart, memberCells;   // cells of cluster i in the order they were absorbed (topological):
                                                 // memberCells[memberStart[i] .. memberStart[i + 1])
    std::vector<int> cLevel, cDepth;          // per cluster: level in the quotient DAG, internal steps
    size_t nClusters() const { return cLevel.size(); }
};

// Clusters are grown IN a topological order of the cell DAG: a cluster only absorbs "ready" cells (all lower
// neighbours placed), preferring the one with most neighbours already inside (ties: the one that became ready
// first); seeds in (dependency level, index) order.  Any such partition has an acyclic quotient graph.
// A cell becomes a candidate when its LAST lower neighbour is placed, so its score (lower neighbours inside the
// cluster being grown) is final at that moment: the candidates sit in one first-in-first-out list per score and a
// pick is the head of the highest non-empty list (round 4; the linear scan over all candidates it replaces was
// 2/3 of the 1.0 s this step took for 10 M cells - same clusters, cell for cell).
inline void cluster_greedy(int nC, int nF, const int* l, const int* u, const int* losort, const int* losortStart,
                           const int* ownerStart, const int* level, int maxCells, ClGreedy& G)
{
    // per-cell state in one record (the walk along the wavefront touches cells in no memory order: one cache line per
    // visit instead of five)
    struct Cell { int indeg, cluster, intra, cnt, cntId; };
    std::vector<Cell> st(nC, Cell{0, -1, 0, 0, -1});
    G.memberStart.assign(1, 0); G.memberCells.clear(); G.memberCells.reserve(nC);
    G.cLevel.clear(); G.cDepth.clear();
    for (int f = 0; f < nF; f++) st[u[f]].indeg++;
    int maxLower = 0;
    for (int c = 0; c < nC; c++) maxLower = std::max(maxLower, losortStart[c + 1] - losortStart[c]);
    // seeds in (dependency level, index) order: the clusters are created along the wavefront, so the
    // fragments left over where blobs do not tile (mesh dimensions that are no multiple of the blob size)
    // depend on their neighbours in parallel instead of forming one serial chain (54^3 box: 51 cluster
    // levels instead of 95 with index-ordered seeds; 40 would be ideal)
    typedef std::pair<int, int> Seed;
    std::priority_queue<Seed, std::vector<Seed>, std::greater<Seed>> ready;
    for (int c = 0; c < nC; c++) if (!st[c].indeg) ready.push(Seed(level[c], c));
    // candidates of the cluster being grown, one FIFO per score
    std::vector<std::vector<int>> bucket(maxLower + 1);
    std::vector<size_t> head(maxLower + 1, 0);
    // (Cell::cnt = lower neighbours already inside the cluster being grown, kept incrementally, valid while cntId == id)
    while (!ready.empty())
    {
        const int seed = ready.top().second; ready.pop();
        if (st[seed].cluster >= 0) continue;
        const int id = (int)G.cLevel.size();
        bucket[0].push_back(seed);
        int top = 0;              // highest score that may have a candidate
        int size = 0, lev = 0, depth = 0;
        while (size < maxCells)
        {
            while (top >= 0 && head[top] == bucket[top].size()) top--;
            if (top < 0) break;
            const int c = bucket[top][head[top]++];
            st[c].cluster = id;
            G.memberCells.push_back(c);
            size++;
            int il = 0;
            for (int j = losortStart[c]; j < losortStart[c + 1]; j++)
            {
                const int p = l[losort[j]];
                if (st[p].cluster == id) il = std::max(il, st[p].intra + 1);
                else lev = std::max(lev, G.cLevel[st[p].cluster] + 1);
            }
            st[c].intra = il;
            depth = std::max(depth, il + 1);
            for (int f = ownerStart[c]; f < ownerStart[c + 1]; f++)
            {
                const int v = u[f];
                Cell& V = st[u[f]];
                if (V.cntId != id) { V.cntId = id; V.cnt = 0; }
                V.cnt++;
                if (--V.indeg == 0)
                {
                    bucket[V.cnt].push_back(v);
                    if (V.cnt > top) top = V.cnt;
                }
            }
        }
        for (int sc = 0; sc <= maxLower; sc++)
        {
            for (size_t t = head[sc]; t < bucket[sc].size(); t++) ready.push(Seed(level[bucket[sc][t]], bucket[sc][t]));
            bucket[sc].clear(); head[sc] = 0;
        }
        G.memberStart.push_back((int)G.memberCells.size());
        G.cLevel.push_back(lev);
        G.cDepth.push_back(depth);
    }
    G.cluster.resize(nC); G.intra.resize(nC);
    for (int c = 0; c < nC; c++) { G.cluster[c] = st[c].cluster; G.intra[c] = st[c].intra; }
}
