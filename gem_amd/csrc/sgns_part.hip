// sgns_part.hip -- the partitioned-table instantiations of sgns_win_kernel (sgns.hpp, PART): TrainModel in walk order restricted to ONE bucket
// (contexts of SynPos partition g, centre words and negatives of SynNeg partition h) of the N-GPU episode schedule (DESIGN.md section 6,
// gemhip_sgns_train_part).  Hogwild (delta write-back + reload-on-update) for the product path, overwrite-on-leave on one wavefront for the parity tests.
#include "sgns.hpp"

namespace gemhip {
sgns_fn pick_sgns_win_part(int d, bool hogwild)
{
    if (d % 2 == 0) {
        const int nv = (d + 127) / 128;
        if (hogwild) return nv <= 1 ? launch_sgns_win_part<2, 1, true> : nv <= 2 ? launch_sgns_win_part<2, 2, true> : nv <= 4 ? launch_sgns_win_part<2, 4, true> : nullptr;
        return nv <= 1 ? launch_sgns_win_part<2, 1, false> : nv <= 2 ? launch_sgns_win_part<2, 2, false> : nv <= 4 ? launch_sgns_win_part<2, 4, false> : nullptr;
    }
    const int nv = (d + 63) / 64;
    if (hogwild) return nv <= 1 ? launch_sgns_win_part<1, 1, true> : nv <= 2 ? launch_sgns_win_part<1, 2, true> : nv <= 4 ? launch_sgns_win_part<1, 4, true> : nullptr;
    return nv <= 1 ? launch_sgns_win_part<1, 1, false> : nv <= 2 ? launch_sgns_win_part<1, 2, false> : nv <= 4 ? launch_sgns_win_part<1, 4, false> : nullptr;
}
}  // namespace gemhip
