/* TOOL (host only, compiled by tools/numbering_probe.py with gcc): dependency statistics of a GaussSeidel sweep on an
 * upper-triangular lduAddressing (faces sorted by owner: GaussSeidelSmoother.C:151-176 walks the rows in cell order, a row
 * needs the NEW values of its lower neighbours and the OLD values of its upper ones).
 *   dag_levels:  level(c) = 1 + max level(lower neighbours)                       -> number of levels (one sweep)
 *   dag_steps_k: row DAG of k pipelined sweeps, T_0 = level, T_j(r) = 1 + max(T_j(lower), T_{j-1}(upper), T_{j-1}(r))
 *                -> max T_{k-1} (what the block engine's per-sweep groupings reach)
 *   greedy_colour: smallest colour not used by an already coloured neighbour, cells taken in the given order
 *   gather_lines: mean number of distinct 128-byte lines (16 doubles) a 64-row wavefront touches when it gathers entry e of
 *                 its rows from a vector stored in `pos` order (pos[c] = position of cell c), rows taken 64 consecutive
 *                 positions at a time - the locality figure of DESIGN section 7d */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void csr(int nC, int nF, const int* l, const int* u, int** startOut, int** nbrOut, int lowerSide)
{
    int* start = (int*)calloc((size_t)nC + 1, sizeof(int));
    for (int f = 0; f < nF; f++) start[(lowerSide ? u[f] : l[f]) + 1]++;
    for (int c = 0; c < nC; c++) start[c + 1] += start[c];
    int* fill = (int*)malloc(sizeof(int) * ((size_t)nC + 1));
    memcpy(fill, start, sizeof(int) * ((size_t)nC + 1));
    int* nbr = (int*)malloc(sizeof(int) * (size_t)(nF > 0 ? nF : 1));
    for (int f = 0; f < nF; f++)
    {
        if (lowerSide) nbr[fill[u[f]]++] = l[f];     /* lower neighbours of u[f] */
        else nbr[fill[l[f]]++] = u[f];               /* upper neighbours of l[f] */
    }
    free(fill);
    *startOut = start; *nbrOut = nbr;
}

int dag_levels(int nC, int nF, const int* l, const int* u, int* level)
{
    int maxL = 0;
    for (int c = 0; c < nC; c++) level[c] = 1;
    /* faces sorted by owner: when row c is reached every face with u[f] == c has l[f] < c, visited already */
    int *ls, *ln;
    csr(nC, nF, l, u, &ls, &ln, 1);
    for (int c = 0; c < nC; c++)
    {
        int m = 0;
        for (int e = ls[c]; e < ls[c + 1]; e++) if (level[ln[e]] > m) m = level[ln[e]];
        level[c] = m + 1;
        if (level[c] > maxL) maxL = level[c];
    }
    free(ls); free(ln);
    return maxL;
}

int dag_steps_k(int nC, int nF, const int* l, const int* u, int k, long* sumT)
{
    int *ls, *ln, *us, *un;
    csr(nC, nF, l, u, &ls, &ln, 1);
    csr(nC, nF, l, u, &us, &un, 0);
    int* prev = (int*)calloc((size_t)nC, sizeof(int));
    int* cur = (int*)calloc((size_t)nC, sizeof(int));
    int maxT = 0;
    for (int j = 0; j < k; j++)
    {
        maxT = 0;
        long s = 0;
        for (int c = 0; c < nC; c++)
        {
            int m = j ? prev[c] : 0;
            for (int e = ls[c]; e < ls[c + 1]; e++) if (cur[ln[e]] > m) m = cur[ln[e]];
            if (j) for (int e = us[c]; e < us[c + 1]; e++) if (prev[un[e]] > m) m = prev[un[e]];
            cur[c] = m + 1;
            if (cur[c] > maxT) maxT = cur[c];
            s += cur[c];
        }
        if (sumT) *sumT = s;
        int* t = prev; prev = cur; cur = t;
    }
    free(prev); free(cur); free(ls); free(ln); free(us); free(un);
    return maxT;
}

int greedy_colour(int nC, int nF, const int* l, const int* u, const int* order, int* colour)
{
    int *ls, *ln, *us, *un;
    csr(nC, nF, l, u, &ls, &ln, 1);
    csr(nC, nF, l, u, &us, &un, 0);
    for (int c = 0; c < nC; c++) colour[c] = -1;
    int nCol = 0;
    for (int i = 0; i < nC; i++)
    {
        const int c = order[i];
        uint64_t used = 0;
        for (int e = ls[c]; e < ls[c + 1]; e++) if (colour[ln[e]] >= 0 && colour[ln[e]] < 64) used |= 1ull << colour[ln[e]];
        for (int e = us[c]; e < us[c + 1]; e++) if (colour[un[e]] >= 0 && colour[un[e]] < 64) used |= 1ull << colour[un[e]];
        int k = 0;
        while (k < 63 && (used >> k) & 1) k++;
        colour[c] = k;
        if (k + 1 > nCol) nCol = k + 1;
    }
    free(ls); free(ln); free(us); free(un);
    return nCol;
}

double gather_lines(int nC, int nF, const int* l, const int* u, const int* pos)
{
    /* rows in storage order: inv[p] = cell at position p */
    int* inv = (int*)malloc(sizeof(int) * (size_t)nC);
    for (int c = 0; c < nC; c++) inv[pos[c]] = c;
    int *ls, *ln, *us, *un;
    csr(nC, nF, l, u, &ls, &ln, 1);
    csr(nC, nF, l, u, &us, &un, 0);
    double lines = 0.0;
    long gathers = 0;
    int tmp[64];
    for (int p0 = 0; p0 < nC; p0 += 64)
    {
        const int n = nC - p0 < 64 ? nC - p0 : 64;
        int width = 0;
        for (int i = 0; i < n; i++)
        {
            const int c = inv[p0 + i];
            const int w = ls[c + 1] - ls[c] + us[c + 1] - us[c];
            if (w > width) width = w;
        }
        for (int e = 0; e < width; e++)
        {
            int m = 0;
            for (int i = 0; i < n; i++)
            {
                const int c = inv[p0 + i];
                const int nl = ls[c + 1] - ls[c];
                int nb = -1;
                if (e < nl) nb = ln[ls[c] + e];
                else if (e - nl < us[c + 1] - us[c]) nb = un[us[c] + e - nl];
                if (nb >= 0) tmp[m++] = pos[nb] >> 4;
            }
            if (!m) continue;
            /* distinct values of tmp[0..m) */
            int d = 0;
            for (int i = 0; i < m; i++)
            {
                int seen = 0;
                for (int j = 0; j < i; j++) if (tmp[j] == tmp[i]) { seen = 1; break; }
                d += !seen;
            }
            lines += d;
            gathers++;
        }
    }
    free(inv); free(ls); free(ln); free(us); free(un);
    return gathers ? lines / (double)gathers : 0.0;
}
