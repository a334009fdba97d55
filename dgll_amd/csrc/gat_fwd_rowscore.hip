// gat_fwd_rowscore.hip -- second-generation GAT pass 0 (forward) in its row-score form: t_j = h_j . a2 formed from the gathered row
// (gat_kernel.hpp, TROW); the kernel template lives in gat_kernel.hpp.
#include "gat_kernel.hpp"

namespace dgll {
bool gat2_launch_0r(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a) {
    return gat2_launch_kind<0, true>(dtype, lpr, nh, grid, s, a, false);
}
}  // namespace dgll
