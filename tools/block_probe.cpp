// critical-path probe of k pipelined GS sweeps under a block partition
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <queue>
#include <cstring>
using namespace std;
int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb");
    int B = atoi(argv[2]); int k = atoi(argv[3]);
    double win = atof(argv[4]), wout = atof(argv[5]);
    int mode = argc > 6 ? atoi(argv[6]) : 0;  // 0 BFS blobs, 1 contiguous ranges
    int hdr[2]; fread(hdr, 4, 2, f); int nC = hdr[0], nF = hdr[1];
    vector<int> l(nF), u(nF); fread(l.data(), 4, nF, f); fread(u.data(), 4, nF, f); fclose(f);
    // CSR of full graph
    vector<int> deg(nC + 1, 0);
    for (int i = 0; i < nF; i++) { deg[l[i] + 1]++; deg[u[i] + 1]++; }
    for (int i = 0; i < nC; i++) deg[i + 1] += deg[i];
    vector<int> adj(2 * (size_t)nF), pos(deg.begin(), deg.end() - 1);
    for (int i = 0; i < nF; i++) { adj[pos[l[i]]++] = u[i]; adj[pos[u[i]]++] = l[i]; }
    // DAG depth
    vector<int> lev(nC, 0); int depth = 0;
    for (int i = 0; i < nF; i++) lev[u[i]] = max(lev[u[i]], lev[l[i]] + 1);  // faces sorted by owner => valid one pass
    for (int c = 0; c < nC; c++) depth = max(depth, lev[c] + 1);
    // blocks
    vector<int> blk(nC, -1); int nB = 0;
    if (mode == 1) { for (int c = 0; c < nC; c++) blk[c] = c / B; nB = (nC + B - 1) / B; }
    else
    {
        vector<int> q;
        for (int s = 0; s < nC; s++)
        {
            if (blk[s] >= 0) continue;
            q.clear(); q.push_back(s); blk[s] = nB; size_t h = 0;
            while (h < q.size() && (int)q.size() < B)
            {
                int c = q[h++];
                for (int e = deg[c]; e < deg[c + 1] && (int)q.size() < B; e++)
                { int n = adj[e]; if (blk[n] < 0) { blk[n] = nB; q.push_back(n); } }
            }
            nB++;
        }
        if (mode == 2)
        {
            // merge small fragments (< B/4) into a neighbouring block (smallest neighbour block)
            vector<int> sz(nB, 0); for (int c = 0; c < nC; c++) sz[blk[c]]++;
            vector<int> target(nB); for (int b = 0; b < nB; b++) target[b] = b;
            vector<int> best(nB, -1);
            for (int c = 0; c < nC; c++) if (sz[blk[c]] < B / 4)
                for (int e = deg[c]; e < deg[c + 1]; e++) { int nb = blk[adj[e]]; if (nb != blk[c] && sz[nb] >= B / 4 && (best[blk[c]] < 0 || sz[nb] < sz[best[blk[c]]])) best[blk[c]] = nb; }
            for (int b = 0; b < nB; b++) if (sz[b] < B / 4 && best[b] >= 0) { target[b] = best[b]; sz[best[b]] += sz[b]; }
            for (int c = 0; c < nC; c++) blk[c] = target[blk[c]];
        }
    }
    vector<int> sz(nB, 0); for (int c = 0; c < nC; c++) sz[blk[c]]++;
    int used = 0, maxsz = 0; for (int b = 0; b < nB; b++) { if (sz[b]) used++; maxsz = max(maxsz, sz[b]); }
    long cut = 0; for (int i = 0; i < nF; i++) cut += blk[l[i]] != blk[u[i]];
    // local depth per block
    vector<int> llev(nC, 0); vector<int> bdepth(nB, 0);
    for (int i = 0; i < nF; i++) if (blk[l[i]] == blk[u[i]]) llev[u[i]] = max(llev[u[i]], llev[l[i]] + 1);
    for (int c = 0; c < nC; c++) bdepth[blk[c]] = max(bdepth[blk[c]], llev[c] + 1);
    double avgd = 0; int maxd = 0; for (int b = 0; b < nB; b++) { avgd += bdepth[b]; maxd = max(maxd, bdepth[b]); }
    {   // ghosts per block
        vector<int> mark(nC,-1); vector<int> ng(nB,0);
        vector<vector<int>> cells(nB); for (int c=0;c<nC;c++) cells[blk[c]].push_back(c);
        int maxslots=0; long tot=0;
        for (int b=0;b<nB;b++){ for(int c:cells[b]) for(int e=deg[c];e<deg[c+1];e++){int n=adj[e]; if(blk[n]!=b&&mark[n]!=b){mark[n]=b;ng[b]++;}} maxslots=max(maxslots,(int)cells[b].size()+ng[b]); tot+=ng[b]; }
        printf("ghosts total %ld (%.2f per cell) maxslots %d | ", tot, (double)tot/nC, maxslots);
    }
    // weighted critical path, k sweeps, row granularity
    vector<double> Tp(nC, 0.0), Tj(nC, 0.0);
    // lower-neighbour lists: faces where c is upper; in face order
    double total = 0;
    vector<double> sweepEnd;
    for (int j = 0; j < k; j++)
    {
        for (int c = 0; c < nC; c++) Tj[c] = j ? Tp[c] : 0.0;
        if (j) for (int i = 0; i < nF; i++) { double w = blk[l[i]] == blk[u[i]] ? win : wout; Tj[l[i]] = max(Tj[l[i]], Tp[u[i]] + w - win); }
        // (a row's own cost win is added when it completes: T = ready + win ; edges from other blocks add wout - win extra)
        for (int i = 0; i < nF; i++)
        {
            // faces sorted by owner: when we reach owner l's faces all its lower deps are final... need T[l] complete before use:
            // T complete(c) = Tj[c] + win, computed lazily: Tj holds ready time; finalize by processing cells in order
        }
        // process cells in order using CSR (neighbors < c are lower)
        for (int c = 0; c < nC; c++)
        {
            double r = Tj[c];
            for (int e = deg[c]; e < deg[c + 1]; e++) { int n = adj[e]; if (n < c) { double w = blk[n] == blk[c] ? 0.0 : wout - win; r = max(r, Tj[n] + w); } }
            Tj[c] = r + win;   // completion time
        }
        double mx = 0; for (int c = 0; c < nC; c++) mx = max(mx, Tj[c]);
        sweepEnd.push_back(mx);
        swap(Tp, Tj);
        total = mx;
    }
    printf("%s nC %d nF %d depth %d | B %d blocks %d maxsz %d cutfrac %.3f localdepth avg %.1f max %d | k %d: ", argv[1], nC, nF, depth, B, used, maxsz, (double)cut / nF, avgd / max(1, used), maxd, k);
    for (double e : sweepEnd) printf("%.0f ", e);
    printf("us  (all-out: %.0f for 1 sweep)\n", depth * wout);
    return 0;
}
