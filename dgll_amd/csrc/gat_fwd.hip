// gat_fwd.hip -- second-generation GAT pass 0 (forward); the kernel template lives in gat_kernel.hpp.
#include "gat_kernel.hpp"

namespace dgll {
bool gat2_launch_0(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow) {
    return gat2_launch_kind<0>(dtype, lpr, nh, grid, s, a, inrow);
}
}  // namespace dgll
