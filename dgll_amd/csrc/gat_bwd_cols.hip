// gat_bwd_cols.hip -- second-generation GAT pass 2 (backward over the rows of A^T); the kernel template lives in gat_kernel.hpp.
#include "gat_kernel.hpp"

namespace dgll {
bool gat2_launch_2(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow) {
    return gat2_launch_kind<2>(dtype, lpr, nh, grid, s, a, inrow);
}
}  // namespace dgll
