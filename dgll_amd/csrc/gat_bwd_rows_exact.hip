// gat_bwd_rows_exact.hip -- second-generation GAT pass 1 (backward over the rows of A) in its exact-dd form ALONE (KIND 3: dd_i from the
// pass's own dot products, no code of the stored-output form); the kernel template lives in gat_kernel.hpp.
#include "gat_kernel.hpp"

namespace dgll {
bool gat2_launch_3(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a, bool inrow) {
    return gat2_launch_kind<3>(dtype, lpr, nh, grid, s, a, inrow);
}
// the same pass with t_j formed from the gathered rows (gat_kernel.hpp, TROW)
bool gat2_launch_3r(int dtype, int lpr, int nh, dim3 grid, hipStream_t s, const EdgeArgs& a) {
    return gat2_launch_kind<3, true>(dtype, lpr, nh, grid, s, a, false);
}
}  // namespace dgll
