// Instantiation unit: conv_h3u_kernel<M_REP> (decoder conv with the operand pass fused into its loader waves, wunet_h3u.h).
#include "wunet_h3u.h"
#include "wunet_launch.h"

#define WUNET_UCASE(M)                                                                                     \
    if (mrep == M) {                                                                                       \
        if (WUNET_ALLOW_BIG_LDS((conv_h3u_kernel<M>), smem) != 0) return -2;                               \
        WUNET_LAUNCH((conv_h3u_kernel<M>), grid, dim3(2 * WUNET_THREADS), smem, st, a);                    \
        return 0;                                                                                          \
    }

int wunet_launch_conv_h3u(const ConvH3uArgs& a, int mrep, dim3 grid, size_t smem, hipStream_t st)
{
    WUNET_UCASE(2) WUNET_UCASE(3) WUNET_UCASE(4)
    return -1;
}
