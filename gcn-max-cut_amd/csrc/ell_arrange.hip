// Host-side preparation of the neighbour table (ELL) the LDS-tiled kernels read.
//
// The gathers of those kernels are LDS-bank bound: a tile row is 64 B = 16 banks, i.e. one of
// four bank "quarters" (row position mod 4), and one ds_read_b128 serves 16 lanes = 4 output
// rows per LDS cycle - conflict-free only if the four neighbour rows it fetches lie in four
// different quarters (random ids: 2.1 cycles on average instead of 1).  The ORDER of a row's
// neighbours in its table slots is free (it only fixes the summation order), and so is the
// quarter of each padding entry (the tile carries four all-zero rows n..n+3).  This routine
// arranges, for every group of four rows that share an LDS cycle, the neighbours over the W
// slots so that each slot's four fetches hit distinct quarters wherever the colour counts allow.
//
// Plain CPU code (data layout, like building the CSR); runs once per batch.
#include "gmc_common.h"

#include <algorithm>
#include <vector>

namespace {

// rows (offsets inside a 16-row block = one wave's rows of a pass, 4 lanes per row) that share
// one LDS cycle of ds_read_b128: lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31},
// {32-35,44-47,52-59}, {36-43,48-51,60-63}  (MI355X_MICROARCH.md, LDS section)
const int kQuad[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

struct Nbr { int id; float w; };

// NS <= W: slots that may hold a neighbour (slots NS..W-1 of every row become padding)
void arrange_quad(const int rows[4], int nrows_valid, const int32_t *rowptr, const int32_t *lcol, const float *vals,
                  int r0, int n, int W, int NS, uint16_t *ell, float *ell_vals) {
    std::vector<Nbr> byc[4][4];  // [row][colour] remaining neighbours (taken from the back)
    int rem[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        if (rows[i] < 0) continue;
        const int r = r0 + rows[i];
        const int end = std::min(rowptr[r + 1], rowptr[r] + W);  // the first W neighbours: the rest are overflow entries
        for (int e = rowptr[r]; e < end; ++e) {
            byc[i][lcol[e] & 3].push_back({lcol[e], vals ? vals[e] : 1.0f});
            ++rem[i];
        }
        for (int c = 0; c < 4; ++c) std::reverse(byc[i][c].begin(), byc[i][c].end());  // keep CSR order per colour
    }
    (void)nrows_valid;
    int perm[4] = {0, 1, 2, 3};
    for (int u = NS; u < W; ++u) {  // the slots the kernels skip
        for (int i = 0; i < 4; ++i) {
            if (rows[i] < 0) continue;
            const long slot = (long)(r0 + rows[i]) * W + u;
            ell[slot] = (uint16_t)(n + (i & 3));
            if (ell_vals) ell_vals[slot] = 0.0f;
        }
    }
    for (int u = 0; u < NS; ++u) {
        const int left = NS - u;  // slots still to fill, this one included
        int D[4] = {0, 0, 0, 0};  // remaining demand per colour
        for (int i = 0; i < 4; ++i)
            for (int c = 0; c < 4; ++c) D[c] += (int)byc[i][c].size();
        int best[4] = {0, 1, 2, 3}, best_conf = 99, best_gain = -1;
        std::sort(perm, perm + 4);
        do {
            int conf = 0, gain = 0;
            for (int i = 0; i < 4; ++i) {
                if (rows[i] < 0 || rem[i] == 0) continue;  // absent row or only padding left
                if (!byc[i][perm[i]].empty()) gain += 4 * D[perm[i]] + (int)byc[i][perm[i]].size();
                else if (rem[i] >= left) ++conf;            // must place a neighbour now: off-colour
            }
            if (conf < best_conf || (conf == best_conf && gain > best_gain)) {
                best_conf = conf; best_gain = gain;
                for (int i = 0; i < 4; ++i) best[i] = perm[i];
            }
        } while (std::next_permutation(perm, perm + 4));
        for (int i = 0; i < 4; ++i) {
            if (rows[i] < 0) continue;
            const long slot = (long)(r0 + rows[i]) * W + u;
            int c = best[i];
            if (rem[i] > 0 && byc[i][c].empty() && rem[i] >= left) {  // forced off-colour: largest stock
                for (int cc = 0; cc < 4; ++cc)
                    if (byc[i][cc].size() > byc[i][c].size()) c = cc;
            }
            if (rem[i] > 0 && !byc[i][c].empty()) {
                const Nbr nb = byc[i][c].back();
                byc[i][c].pop_back();
                --rem[i];
                ell[slot] = (uint16_t)nb.id;
                if (ell_vals) ell_vals[slot] = nb.w;
            } else {  // padding: the all-zero row of the wanted quarter
                ell[slot] = (uint16_t)(n + ((best[i] - n) & 3));
                if (ell_vals) ell_vals[slot] = 0.0f;
            }
        }
    }
    // Refinement.  The slot-by-slot pass above leaves its conflicts to the last slots; without padding to
    // absorb them (7 live slots, every row full) they pile up there.  Hill-climb over swaps of two slots
    // inside one row: only those two slots' costs change; cost of a slot = LDS cycles of its fetch = the
    // largest number of DISTINCT rows of one quarter among the group's ids.
    auto slot_cost = [&](int u) {
        int ids[4], k = 0, cnt[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            if (rows[i] < 0) continue;
            const int id = ell[(long)(r0 + rows[i]) * W + u];
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= ids[j] == id;   // identical addresses broadcast
            if (!dup) { ids[k++] = id; ++cnt[id & 3]; }
        }
        return std::max(std::max(cnt[0], cnt[1]), std::max(cnt[2], cnt[3]));
    };
    for (int pass = 0, improved = 1; pass < 8 && improved; ++pass) {
        improved = 0;
        for (int i = 0; i < 4; ++i) {
            if (rows[i] < 0) continue;
            const long base = (long)(r0 + rows[i]) * W;
            for (int u = 0; u < NS; ++u)
                for (int v = u + 1; v < NS; ++v) {
                    if ((ell[base + u] & 3) == (ell[base + v] & 3)) continue;  // same quarter: nothing changes
                    const int before = slot_cost(u) + slot_cost(v);
                    std::swap(ell[base + u], ell[base + v]);
                    if (ell_vals) std::swap(ell_vals[base + u], ell_vals[base + v]);
                    if (slot_cost(u) + slot_cost(v) < before) { improved = 1; continue; }
                    std::swap(ell[base + u], ell[base + v]);
                    if (ell_vals) std::swap(ell_vals[base + u], ell_vals[base + v]);
                }
        }
    }
}

}  // namespace

// All pointers are HOST pointers.  ell: [R][W] (W = 8 or 16; a longer row's first W neighbours), ell_vals: [R][W] or NULL.
// Padding entries are n_g .. n_g+3 (four all-zero tile rows, one per bank quarter).
extern "C" int gmc_ell_slots_for(int32_t R, const int32_t *rowptr, int32_t W) {
    if ((W != 8 && W != 16) || !rowptr) return W;
    int top = 0;
    for (int r = 0; r < R; ++r) top = std::max(top, std::min(rowptr[r + 1] - rowptr[r], W));
    return std::max(top, W == 8 ? 7 : 9);   // the kernels' smallest specialisations
}

extern "C" int gmc_ell_arrange_host(int32_t B, const int32_t *goff, const int32_t *rowptr, const int32_t *lcol,
                                    const float *vals, int32_t W, uint16_t *ell, float *ell_vals) {
    if (!goff || !rowptr || !lcol || !ell) return GMC_ERR_NULL;
    if (B < 0 || (W != 8 && W != 16)) return GMC_ERR_SHAPE;
    const int NS = B > 0 ? gmc_ell_slots_for(goff[B], rowptr, W) : W;
    for (int g = 0; g < B; ++g) {
        const int r0 = goff[g], n = goff[g + 1] - r0;
        if (n + 4 > 65535) return GMC_ERR_GRAPH_SIZE;
        for (int blk = 0; blk < n; blk += 16) {
            for (int qd = 0; qd < 4; ++qd) {
                int rows[4], valid = 0;
                for (int i = 0; i < 4; ++i) {
                    const int l = blk + kQuad[qd][i];
                    rows[i] = l < n ? l : -1;
                    valid += l < n;
                }
                if (valid) arrange_quad(rows, valid, rowptr, lcol, vals, r0, n, W, NS, ell, ell_vals);
            }
        }
    }
    return GMC_OK;
}
