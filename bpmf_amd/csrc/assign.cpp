// assign.cpp -- work-balanced assignment of the columns of a side to the ranks (GPUs).
//
// bpmf_assign_greedy restates Sys::assign of the reference (c++/assign.cpp:52-201) for its default settings: every
// column goes, in index order, to the rank whose share of the work assigned so far is smallest (work of a column =
// 10 + nnz; the communication-cost term has weight 0 in the reference, :158), the sweep is repeated three times with
// every column first taken out again (with the reference's 7.1 + nnz on the way out, :118), and the columns are then
// renumbered so that every rank owns a contiguous range (the permutation of :185-190, each rank's columns in
// ascending original order).  bpmf_assign_contiguous cuts the ORIGINAL order at equal c0 + nnz instead (no
// permutation: column ids, hence the per-column RNG streams and the samples, do not depend on the number of ranks).
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "../../include/bpmf_io.h"

extern "C" BPMF_IO_API int bpmf_assign_greedy(int64_t n, const int64_t *colptr, int nparts, int64_t *order, int64_t *dom)
{
    if (n < 0 || !colptr || nparts < 1 || !order || !dom) return -1;
    std::vector<double> work((size_t)nparts, 0.0);
    std::vector<int64_t> count((size_t)nparts, 0);
    std::vector<int> owner((size_t)n, -1);
    double total = 0.01;                                             // (total_work starts at 0.01, :88)
    for (int pass = 0; pass < 3; ++pass) {
        for (int64_t i = 0; i < n; ++i) {
            const double nnz = (double)(colptr[i + 1] - colptr[i]);
            if (owner[(size_t)i] >= 0) {                             // unassign (:113-126)
                const int p = owner[(size_t)i];
                work[(size_t)p] -= 7.1 + nnz; total -= 7.1 + nnz; count[(size_t)p]--;
                owner[(size_t)i] = -1;
            }
            int best = -1;                                           // best (:92-106): ties go to the LAST rank with the minimum
            double min_cost = 1e9;
            for (int p = 0; p < nparts; ++p) {
                const double cost = 10000.0 * (work[(size_t)p] / total);
                if (cost > min_cost) continue;
                best = p; min_cost = cost;
            }
            if (best < 0) best = nparts - 1;
            owner[(size_t)i] = best;                                 // assign (:109-120)
            work[(size_t)best] += 10.0 + nnz; total += 10.0 + nnz; count[(size_t)best]++;
        }
    }
    dom[0] = 0;
    for (int p = 0; p < nparts; ++p) dom[p + 1] = dom[p] + count[(size_t)p];
    std::vector<int64_t> pos(dom, dom + nparts);
    for (int64_t i = 0; i < n; ++i) order[pos[(size_t)owner[(size_t)i]]++] = i;     // new position -> old column
    return 0;
}

extern "C" BPMF_IO_API int bpmf_assign_contiguous(int64_t n, const int64_t *colptr, int nparts, double c0, int64_t *dom)
{
    if (n < 0 || !colptr || nparts < 1 || !dom) return -1;
    const double total = (double)(colptr[n] - colptr[0]) + c0 * (double)n;
    int64_t col = 0;
    dom[0] = 0;
    for (int p = 1; p < nparts; ++p) {
        const double goal = total * p / nparts;
        while (col < n && (double)(colptr[col + 1] - colptr[0]) + c0 * (double)(col + 1) <= goal) ++col;
        dom[p] = col;
    }
    dom[nparts] = n;
    return 0;
}
