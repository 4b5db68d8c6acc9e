// sgns_det.hip -- the deterministic instantiations (sgns.hpp): sgns_win_kernel with overwrite on leave and sgns_kernel (no LDS window).
// One wavefront of either reproduces oracle/n2v_oracle.c to 2e-4; sgns_kernel is also the Hogwild fallback for d >= 384 (window does not fit).
#include "sgns.hpp"

namespace gemhip {
sgns_fn pick_sgns_win_det(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        return nv <= 1 ? launch_sgns_win<2, 1, false> : nv <= 2 ? launch_sgns_win<2, 2, false> : nv <= 4 ? launch_sgns_win<2, 4, false> : nullptr;
    }
    const int nv = (d + 63) / 64;
    return nv <= 1 ? launch_sgns_win<1, 1, false> : nv <= 2 ? launch_sgns_win<1, 2, false> : nv <= 4 ? launch_sgns_win<1, 4, false> : nullptr;
}

sgns_fn pick_sgns(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        if (nv <= 1) return launch_sgns<2, 1>;
        if (nv <= 2) return launch_sgns<2, 2>;
        if (nv <= 4) return launch_sgns<2, 4>;
        return nullptr;
    }
    const int nv = (d + 63) / 64;
    if (nv <= 1) return launch_sgns<1, 1>;
    if (nv <= 2) return launch_sgns<1, 2>;
    if (nv <= 4) return launch_sgns<1, 4>;
    return nullptr;
}
}  // namespace gemhip
