// Instantiation unit: wgrad_mfma_kernel<WUNET_INST_TAPS, mrep, NW, XIT, WSPLIT> for every tiling the
// planner (plan_wgrad in wunet_api.cpp) can choose.  Compiled once per tap count by the Makefile.
#include "wunet_launch.h"

#define WUNET_CAT2(a, b) wunet_launch_wgrad_##a
#define WUNET_CAT(a) WUNET_CAT2(a, )
#define WUNET_CASE(M, NW_, XIT_, WS_)                                                                             \
    if (mrep == M && nw == NW_ && xit == XIT_ && wsplit == (WS_ ? 1 : 0)) {                                        \
        WUNET_LAUNCH((wgrad_mfma_kernel<WUNET_INST_TAPS, M, NW_, XIT_, WS_>), grid, dim3(WUNET_THREADS), smem, st, a); \
        return 0;                                                                                                  \
    }
#define WUNET_CASES(NW_, XIT_, WS_) WUNET_CASE(2, NW_, XIT_, WS_) WUNET_CASE(3, NW_, XIT_, WS_) WUNET_CASE(4, NW_, XIT_, WS_) \
    WUNET_CASE(5, NW_, XIT_, WS_) WUNET_CASE(6, NW_, XIT_, WS_)

int WUNET_CAT(WUNET_INST_TAPS)(const WgradArgs& a, int mrep, int nw, int xit, int wsplit, dim3 grid, size_t smem, hipStream_t st)
{
#if WUNET_INST_TAPS == 15
    WUNET_CASES(6, 2, false)
    WUNET_CASES(6, 8, false)
    WUNET_CASES(1, 1, true)
#else
    WUNET_CASES(2, 2, false)
    WUNET_CASES(2, 8, false)
    WUNET_CASES(6, 6, false)
#endif
    return -1;
}
