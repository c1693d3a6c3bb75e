// bigfused_4096.hip -- instantiates the workgroup-level fused kernel for N = 4096 (see bigfused_impl.h)
#include "bigfused_impl.h"

namespace psdk {

hipError_t launch_bigfused_4096(const FusedBatch &b, const float *win, const cf *tw0g, const cf *twag, hipStream_t s,
                                 hipEvent_t ea, hipEvent_t eb)
{
    return launch_bigfused_n<4096>(b, win, tw0g, twag, s, ea, eb);
}

} // namespace psdk
