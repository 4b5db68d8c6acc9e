// sgns_hogwild.hip -- the Hogwild instantiations of sgns_win_kernel (sgns.hpp): one wavefront per walk, LDS window with delta write-back,
// reload-on-update of the negative rows, hot rows kept out of the window.  What gemhip_sgns_train launches unless flags|4 (deterministic).
#include "sgns.hpp"

namespace gemhip {
sgns_fn pick_sgns_win_hogwild(int d)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        return nv <= 1 ? launch_sgns_win<2, 1, true> : nv <= 2 ? launch_sgns_win<2, 2, true> : nv <= 4 ? launch_sgns_win<2, 4, true> : nullptr;
    }
    const int nv = (d + 63) / 64;
    return nv <= 1 ? launch_sgns_win<1, 1, true> : nv <= 2 ? launch_sgns_win<1, 2, true> : nv <= 4 ? launch_sgns_win<1, 4, true> : nullptr;
}
}  // namespace gemhip
