// graphlily/app/row_shards.h -- one process per GPU: which rows of the matrix this rank owns.
//
// No counterpart in the reference (single device).  The app drivers of this directory cut the matrix into contiguous,
// nnz-balanced row ranges whose interior boundaries are multiples of 64 rows (whole 64-bit words of a bit vector, 256-byte
// aligned slices of a float vector) -- the rule of graphlily_amd/dist.py:partition_rows_by_nnz, so that C++ and Python ranks
// agree -- and exchange through the C ABI's gl_dist_* calls (grouped RCCL sends / receives on the library's stream, recordable
// into a hipGraph).  Without set_comm() a driver is a world of one and none of this does anything.
#ifndef GRAPHLILY_HIP_APP_ROW_SHARDS_H_
#define GRAPHLILY_HIP_APP_ROW_SHARDS_H_

#include <algorithm>
#include <cstdint>
#include <vector>

#include "graphlily/global.h"

namespace graphlily {
namespace app {

struct RowShards {
    gl_dist comm = nullptr;          // not owned
    int rank = 0, world = 1;
    std::vector<uint32_t> bounds;    // world + 1 row boundaries; rank r owns [bounds[r], bounds[r + 1])

    void set_comm(gl_dist c) {
        comm = c;
        rank = 0;
        world = 1;
        if (c) GRAPHLILY_CHECK(gl_dist_rank(c, &rank, &world));
    }
    uint32_t row_begin() const { return bounds[rank]; }
    uint32_t row_end() const { return bounds[rank + 1]; }

    // boundaries from a CSR row pointer: equal row counts when those are already balanced to 3 % (randomly labelled graphs:
    // equal slices need no padding in the all-gather), else the row whose prefix reaches k / world of the non-zeros, rounded
    // to the nearest multiple of `align`
    template <typename IndPtr>
    void cut(const IndPtr &indptr, uint32_t align = 64) {
        const uint32_t n = (uint32_t)indptr.size() - 1;
        const uint64_t nnz = indptr[n];
        bounds.assign((size_t)world + 1, 0);
        bounds[world] = n;
        if (world == 1) return;
        if (n % ((uint32_t)world * align) == 0 && nnz > 0) {
            uint64_t most = 0;
            for (int r = 0; r < world; r++) most = std::max<uint64_t>(most, (uint64_t)indptr[n / world * (r + 1)] - indptr[n / world * r]);
            if ((double)most <= 1.03 * (double)nnz / world) {
                for (int r = 1; r < world; r++) bounds[r] = n / world * r;
                return;
            }
        }
        for (int k = 1; k < world; k++) {
            const uint64_t target = nnz * (uint64_t)k / (uint64_t)world;
            uint32_t r = (uint32_t)(std::lower_bound(indptr.begin(), indptr.end(), target) - indptr.begin());
            r = (uint32_t)(((uint64_t)r + align / 2) / align * align);
            bounds[k] = std::min(n, std::max(bounds[k - 1], r));
        }
    }
};

}  // namespace app
}  // namespace graphlily

#endif  // GRAPHLILY_HIP_APP_ROW_SHARDS_H_
