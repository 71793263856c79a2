// Instantiation unit: wgrad_h3_kernel<TAPS, M_REP, NSEG, TP> and wgrad_h3d_kernel<TAPS, M_REP, DB> (fp16-split weight gradients).
#include "wunet_h3.h"
#include "wunet_launch.h"

#define WUNET_WCASE(T, M, S, P)                                                                            \
    if (taps == T && mrep == M && nseg == S && tp == P) {                                                  \
        if (bf) {                                                                                          \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3_kernel<T, M, S, P, true>), smem) != 0) return -2;            \
            WUNET_LAUNCH((wgrad_h3_kernel<T, M, S, P, true>), grid, dim3(WUNET_THREADS), smem, st, a);     \
        } else {                                                                                           \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3_kernel<T, M, S, P>), smem) != 0) return -2;                  \
            WUNET_LAUNCH((wgrad_h3_kernel<T, M, S, P>), grid, dim3(WUNET_THREADS), smem, st, a);           \
        }                                                                                                  \
        return 0;                                                                                          \
    }
#define WUNET_WCASES(T, M) WUNET_WCASE(T, M, 1, 128) WUNET_WCASE(T, M, 2, 128) WUNET_WCASE(T, M, 4, 128) WUNET_WCASE(T, M, 8, 128)

int wunet_launch_wgrad_h3(const WgradH3Args& a, int taps, int mrep, int nseg, int tp, dim3 grid, size_t smem, hipStream_t st, bool bf)
{
    WUNET_WCASES(15, 2) WUNET_WCASES(15, 3) WUNET_WCASES(15, 4) WUNET_WCASES(15, 5) WUNET_WCASES(15, 6)
    WUNET_WCASES(5, 2) WUNET_WCASES(5, 3) WUNET_WCASES(5, 4) WUNET_WCASES(5, 5) WUNET_WCASES(5, 6)
    return -1;
}

#define WUNET_DCASE(T, M, D)                                                                               \
    if (taps == T && mrep == M && db == D) {                                                               \
        if (bf) {                                                                                          \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3d_kernel<T, M, D, true>), smem) != 0) return -2;              \
            WUNET_LAUNCH((wgrad_h3d_kernel<T, M, D, true>), grid, dim3(WUNET_THREADS), smem, st, a);       \
        } else {                                                                                           \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3d_kernel<T, M, D>), smem) != 0) return -2;                    \
            WUNET_LAUNCH((wgrad_h3d_kernel<T, M, D>), grid, dim3(WUNET_THREADS), smem, st, a);             \
        }                                                                                                  \
        return 0;                                                                                          \
    }

int wunet_launch_wgrad_h3d(const WgradH3dArgs& a, int taps, int mrep, bool db, dim3 grid, size_t smem, hipStream_t st, bool bf, int tp)
{
    if (tp != 128) return -1;
    WUNET_DCASE(15, 2, true) WUNET_DCASE(15, 3, true) WUNET_DCASE(15, 4, true) WUNET_DCASE(15, 5, true) WUNET_DCASE(15, 6, true)
    WUNET_DCASE(5, 2, true) WUNET_DCASE(5, 3, true) WUNET_DCASE(5, 4, true) WUNET_DCASE(5, 5, true)
    WUNET_DCASE(15, 2, false) WUNET_DCASE(15, 3, false) WUNET_DCASE(5, 2, false) WUNET_DCASE(5, 3, false) WUNET_DCASE(5, 4, false)
    return -1;
}
