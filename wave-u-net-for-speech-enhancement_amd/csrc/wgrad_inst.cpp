// Instantiation unit: wgrad_mfma_kernel<WUNET_INST_TAPS, WUNET_INST_MODE, mrep, NW>.
#include "wunet_launch.h"

#define WUNET_CAT2(a, b, c) wunet_launch_wgrad_##a##_##b
#define WUNET_CAT(a, b) WUNET_CAT2(a, b, )
#define WUNET_NW ((WUNET_INST_TAPS) == 15 ? 6 : 2)
#define WUNET_CASE(M)                                                                                          \
    if (mrep == M) {                                                                                           \
        WUNET_LAUNCH((wgrad_mfma_kernel<WUNET_INST_TAPS, WUNET_INST_MODE, M, WUNET_NW>), grid, dim3(WUNET_THREADS), smem, st, a); \
        return 0;                                                                                              \
    }

int WUNET_CAT(WUNET_INST_TAPS, WUNET_INST_MODE)(const WgradArgs& a, int mrep, dim3 grid, size_t smem, hipStream_t st)
{
    WUNET_CASE(2) WUNET_CASE(3) WUNET_CASE(4) WUNET_CASE(5) WUNET_CASE(6)
    return -1;
}
