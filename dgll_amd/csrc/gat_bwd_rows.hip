// gat_bwd_rows.hip -- second-generation GAT pass 1 (backward over the rows of A); the kernel template lives in gat_kernel.hpp.
#include "gat_kernel.hpp"

namespace dgll {
bool gat2_launch_1(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow) {
    return gat2_launch_kind<1>(dtype, lpr, nh, grid, s, a, inrow);
}
}  // namespace dgll
