// grouping experiment: tasks = (sweep, group) where group = rows of one block with equal potential phi (chunks of 64);
// group-level DAG times for k sweeps; phi = level (T0) or row-level T1 / T2
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
using namespace std;
int nC, nF; vector<int> l, u, deg, adj;
static vector<int> rowT(int j, const vector<int>& Tp) {   // row-level unweighted T_j from T_{j-1}
    vector<int> T(nC, 0);
    if (j) { for (int c = 0; c < nC; c++) T[c] = Tp[c]; for (int i = 0; i < nF; i++) T[l[i]] = max(T[l[i]], Tp[u[i]]); }
    for (int c = 0; c < nC; c++) { int r = T[c]; for (int e = deg[c]; e < deg[c + 1]; e++) { int n = adj[e]; if (n < c) r = max(r, T[n]); } T[c] = r + 1; }
    return T;
}
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); int B = atoi(argv[2]); int k = atoi(argv[3]); int which = atoi(argv[4]);
    int hdr[2]; fread(hdr, 4, 2, f); nC = hdr[0]; nF = hdr[1];
    l.resize(nF); u.resize(nF); fread(l.data(), 4, nF, f); fread(u.data(), 4, nF, f); fclose(f);
    deg.assign(nC + 1, 0);
    for (int i = 0; i < nF; i++) { deg[l[i] + 1]++; deg[u[i] + 1]++; }
    for (int i = 0; i < nC; i++) deg[i + 1] += deg[i];
    adj.resize(2 * (size_t)nF); { vector<int> pos(deg.begin(), deg.end() - 1); for (int i = 0; i < nF; i++) { adj[pos[l[i]]++] = u[i]; adj[pos[u[i]]++] = l[i]; } }
    // blocks (BFS blobs + merge)
    vector<int> blk(nC, -1); int nB = 0;
    { vector<int> q; for (int s = 0; s < nC; s++) { if (blk[s] >= 0) continue; q.clear(); q.push_back(s); blk[s] = nB; size_t h = 0;
        while (h < q.size() && (int)q.size() < B) { int c = q[h++]; for (int e = deg[c]; e < deg[c + 1] && (int)q.size() < B; e++) { int n = adj[e]; if (blk[n] < 0) { blk[n] = nB; q.push_back(n); } } } nB++; }
      vector<int> sz(nB, 0); for (int c = 0; c < nC; c++) sz[blk[c]]++; vector<int> best(nB, -1), target(nB);
      for (int b = 0; b < nB; b++) target[b] = b;
      for (int c = 0; c < nC; c++) if (sz[blk[c]] < B / 4) for (int e = deg[c]; e < deg[c + 1]; e++) { int nb = blk[adj[e]]; if (nb != blk[c] && sz[nb] >= B / 4 && (best[blk[c]] < 0 || sz[nb] < sz[best[blk[c]]])) best[blk[c]] = nb; }
      for (int b = 0; b < nB; b++) if (sz[b] < B / 4 && best[b] >= 0) { target[b] = best[b]; sz[best[b]] += sz[b]; }
      for (int c = 0; c < nC; c++) blk[c] = target[blk[c]]; }
    vector<vector<int>> RT(4); RT[0] = rowT(0, RT[0]); for (int j = 1; j < 4; j++) RT[j] = rowT(j, RT[j - 1]);
    printf("%s nC %d: row-level T max per sweep: %d %d %d %d\n", argv[1], nC, *max_element(RT[0].begin(), RT[0].end()), *max_element(RT[1].begin(), RT[1].end()), *max_element(RT[2].begin(), RT[2].end()), *max_element(RT[3].begin(), RT[3].end()));
    // groups: which = 0: phi = T0 for every sweep; 1: phi = T1 for every sweep; 2: T0 for sweep 0, T1 for sweeps >= 1; 3: T_j for sweep j
    auto grouping = [&](const vector<int>& phi, vector<int>& grp) -> int {
        map<pair<int,int>, pair<int,int>> cur;  // (blk, phi) -> (group id, count)
        int nG = 0; grp.assign(nC, 0);
        for (int c = 0; c < nC; c++) { auto key = make_pair(blk[c], phi[c]); auto it = cur.find(key); if (it == cur.end() || it->second.second == 64) { cur[key] = make_pair(nG++, 1); grp[c] = nG - 1; } else { it->second.second++; grp[c] = it->second.first; } }
        return nG;
    };
    vector<vector<int>> G(k); vector<int> nG(k);
    for (int j = 0; j < k; j++) { const vector<int>& phi = which == 0 ? RT[0] : which == 1 ? RT[1] : which == 2 ? RT[min(j, 1)] : RT[j]; nG[j] = grouping(phi, G[j]); }
    // group-level times: process rows in an order consistent with phi of that sweep (sort by phi)
    vector<int> Tprev; long totalTasks = 0; 
    for (int j = 0; j < k; j++)
    {
        const vector<int>& phi = which == 0 ? RT[0] : which == 1 ? RT[1] : which == 2 ? RT[min(j, 1)] : RT[j];
        vector<int> T(nG[j], 0);
        if (j) { for (int c = 0; c < nC; c++) T[G[j][c]] = max(T[G[j][c]], Tprev[G[j - 1][c]]); for (int i = 0; i < nF; i++) T[G[j][l[i]]] = max(T[G[j][l[i]]], Tprev[G[j - 1][u[i]]]); }
        vector<int> order(nC); for (int c = 0; c < nC; c++) order[c] = c; stable_sort(order.begin(), order.end(), [&](int a, int b) { return phi[a] < phi[b]; });
        // groups in phi order: all rows of lower phi final before; group time = 1 + max over rows' lower nbr groups
        // two-phase per phi value: first gather max, then +1 (rows with equal phi are independent)
        size_t i0 = 0;
        while (i0 < order.size()) { size_t i1 = i0; while (i1 < order.size() && phi[order[i1]] == phi[order[i0]]) i1++;
            for (size_t i = i0; i < i1; i++) { int c = order[i]; int g = G[j][c]; for (int e = deg[c]; e < deg[c + 1]; e++) { int n = adj[e]; if (n < c) T[g] = max(T[g], T[G[j][n]] ); } }
            // mark +1 once per group: use a flag
            for (size_t i = i0; i < i1; i++) { int g = G[j][order[i]]; if (T[g] >= 0) T[g] = -(T[g] + 1) - 1; }   // encode done: negative
            for (size_t i = i0; i < i1; i++) { int g = G[j][order[i]]; if (T[g] < 0) T[g] = -(T[g] + 1); }
            i0 = i1; }
        printf("  sweep %d: %d groups (%.1f rows each), group-level T max %d\n", j, nG[j], (double)nC / nG[j], *max_element(T.begin(), T.end()));
        totalTasks += nG[j]; Tprev = T;
    }
    return 0;
}
