// Instantiation unit: conv_mfma_kernel<WUNET_INST_TAPS, mrep, nrep> for every (mrep, nrep) the planner
// can choose.  Compiled once per tap count by the Makefile.
#include "wunet_launch.h"

#define WUNET_CAT2(a, b) wunet_launch_conv_##a
#define WUNET_CAT(a) WUNET_CAT2(a, )
#define WUNET_CASE(M, N)                                                                                       \
    if (mrep == M && nrep == N) {                                                                              \
        if (WUNET_ALLOW_BIG_LDS((conv_mfma_kernel<WUNET_INST_TAPS, M, N>), smem) != 0) return -2;              \
        WUNET_LAUNCH((conv_mfma_kernel<WUNET_INST_TAPS, M, N>), grid, dim3(WUNET_THREADS), smem, st, a);       \
        return 0;                                                                                              \
    }

int WUNET_CAT(WUNET_INST_TAPS)(const ConvArgs& a, int mrep, int nrep, dim3 grid, size_t smem, hipStream_t st)
{
    WUNET_CASE(2, 1) WUNET_CASE(3, 1) WUNET_CASE(4, 1) WUNET_CASE(5, 1) WUNET_CASE(6, 1)
    WUNET_CASE(2, 4) WUNET_CASE(3, 4) WUNET_CASE(4, 4) WUNET_CASE(5, 4) WUNET_CASE(6, 4)
    return -1;
}
