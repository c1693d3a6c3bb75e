// bigfused3_2048.hip -- instantiates the three-pass workgroup-level fused kernel for N = 2048 (see bigfused3_impl.h)
#include "bigfused3_impl.h"

namespace psdk {

hipError_t launch_bigfused3_2048(const FusedBatch &b, const float *win, const cf *tw3g, hipStream_t s, hipEvent_t ea, hipEvent_t eb)
{
    return launch_bigfused3_n<2048>(b, win, tw3g, s, ea, eb);
}

} // namespace psdk
