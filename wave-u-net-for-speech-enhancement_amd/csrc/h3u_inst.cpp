// Instantiation unit: conv_h3u_kernel<M_REP, COPY> (decoder conv with the operand pass fused into its loader waves, wunet_h3u.h).
#include "wunet_h3u.h"
#include "wunet_launch.h"

// (the training launches - ConvH3uArgs::oxh set: the loaders copy the operand to HBM - run the COPY instantiation)
#define WUNET_UCASE(M)                                                                                     \
    if (mrep == M && a.oxh != nullptr) {                                                                   \
        if (WUNET_ALLOW_BIG_LDS((conv_h3u_kernel<M, true>), smem) != 0) return -2;                         \
        WUNET_LAUNCH((conv_h3u_kernel<M, true>), grid, dim3(2 * WUNET_THREADS), smem, st, a);              \
        return 0;                                                                                          \
    }                                                                                                      \
    if (mrep == M) {                                                                                       \
        if (WUNET_ALLOW_BIG_LDS((conv_h3u_kernel<M, false>), smem) != 0) return -2;                        \
        WUNET_LAUNCH((conv_h3u_kernel<M, false>), grid, dim3(2 * WUNET_THREADS), smem, st, a);             \
        return 0;                                                                                          \
    }

int wunet_launch_conv_h3u(const ConvH3uArgs& a, int mrep, dim3 grid, size_t smem, hipStream_t st)
{
    WUNET_UCASE(2) WUNET_UCASE(3) WUNET_UCASE(4)
    return -1;
}
