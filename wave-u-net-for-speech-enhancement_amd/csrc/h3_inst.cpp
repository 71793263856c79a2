// Instantiation unit: conv_h3_kernel<TAPS, M_REP, NSEG> and wgrad_h3_kernel<TAPS, M_REP, NSEG, TP> (fp16-split GEMM path).
#include "wunet_h3.h"
#include "wunet_launch.h"

#define WUNET_CASE(T, M, S)                                                                                \
    if (taps == T && mrep == M && nseg == S) {                                                             \
        if (bf) {                                                                                          \
            if (WUNET_ALLOW_BIG_LDS((conv_h3_kernel<T, M, S, true>), smem) != 0) return -2;                \
            WUNET_LAUNCH((conv_h3_kernel<T, M, S, true>), grid, dim3(WUNET_THREADS), smem, st, a);         \
        } else {                                                                                           \
            if (WUNET_ALLOW_BIG_LDS((conv_h3_kernel<T, M, S>), smem) != 0) return -2;                      \
            WUNET_LAUNCH((conv_h3_kernel<T, M, S>), grid, dim3(WUNET_THREADS), smem, st, a);               \
        }                                                                                                  \
        return 0;                                                                                          \
    }

int wunet_launch_conv_h3(const ConvH3Args& a, int taps, int mrep, int nseg, dim3 grid, size_t smem, hipStream_t st, bool bf)
{
    WUNET_CASE(15, 2, 1) WUNET_CASE(15, 3, 1) WUNET_CASE(15, 4, 1)
    WUNET_CASE(5, 2, 1) WUNET_CASE(5, 3, 1) WUNET_CASE(5, 4, 1)
    WUNET_CASE(15, 2, 2) WUNET_CASE(15, 3, 2) WUNET_CASE(5, 2, 2) WUNET_CASE(5, 3, 2)
    WUNET_CASE(15, 2, 4) WUNET_CASE(15, 3, 4) WUNET_CASE(5, 2, 4) WUNET_CASE(5, 3, 4)
    WUNET_CASE(15, 2, 8) WUNET_CASE(15, 3, 8) WUNET_CASE(5, 2, 8) WUNET_CASE(5, 3, 8)
    WUNET_CASE(15, 2, 16) WUNET_CASE(15, 3, 16) WUNET_CASE(5, 2, 16) WUNET_CASE(5, 3, 16)
    return -1;
}

#define WUNET_PCASE(T, M)                                                                                  \
    if (taps == T && mrep == M) {                                                                          \
        if (WUNET_ALLOW_BIG_LDS((conv_h3p_kernel<T, M>), smem) != 0) return -2;                            \
        WUNET_LAUNCH((conv_h3p_kernel<T, M>), grid, dim3(2 * WUNET_THREADS), smem, st, a);                 \
        return 0;                                                                                          \
    }

int wunet_launch_conv_h3p(const ConvH3Args& a, int taps, int mrep, dim3 grid, size_t smem, hipStream_t st)
{
    WUNET_PCASE(15, 2) WUNET_PCASE(15, 3) WUNET_PCASE(15, 4) WUNET_PCASE(5, 2) WUNET_PCASE(5, 3) WUNET_PCASE(5, 4)
    return -1;
}

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
#define WUNET_WCASES(T, M) WUNET_WCASE(T, M, 1, 128) WUNET_WCASE(T, M, 2, 128) WUNET_WCASE(T, M, 4, 128) WUNET_WCASE(T, M, 8, 128) \
                           WUNET_WCASE(T, M, 1, 256) WUNET_WCASE(T, M, 2, 256) WUNET_WCASE(T, M, 4, 256)

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

#define WUNET_D64CASE(T, M)                                                                                \
    if (tp == 64 && taps == T && mrep == M && db) {                                                        \
        if (bf) {                                                                                          \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3d_kernel<T, M, true, true, 64>), smem) != 0) return -2;       \
            WUNET_LAUNCH((wgrad_h3d_kernel<T, M, true, true, 64>), grid, dim3(WUNET_THREADS), smem, st, a); \
        } else {                                                                                           \
            if (WUNET_ALLOW_BIG_LDS((wgrad_h3d_kernel<T, M, true, false, 64>), smem) != 0) return -2;      \
            WUNET_LAUNCH((wgrad_h3d_kernel<T, M, true, false, 64>), grid, dim3(WUNET_THREADS), smem, st, a); \
        }                                                                                                  \
        return 0;                                                                                          \
    }

int wunet_launch_wgrad_h3d(const WgradH3dArgs& a, int taps, int mrep, bool db, dim3 grid, size_t smem, hipStream_t st, bool bf, int tp)
{
    WUNET_D64CASE(15, 2) WUNET_D64CASE(15, 3) WUNET_D64CASE(5, 2) WUNET_D64CASE(5, 3) WUNET_D64CASE(5, 4)
    if (tp != 128) return -1;
    WUNET_DCASE(15, 2, true) WUNET_DCASE(15, 3, true) WUNET_DCASE(15, 4, true) WUNET_DCASE(15, 5, true) WUNET_DCASE(15, 6, true)
    WUNET_DCASE(5, 2, true) WUNET_DCASE(5, 3, true) WUNET_DCASE(5, 4, true) WUNET_DCASE(5, 5, true)
    WUNET_DCASE(15, 2, false) WUNET_DCASE(15, 3, false) WUNET_DCASE(5, 2, false) WUNET_DCASE(5, 3, false) WUNET_DCASE(5, 4, false)
    return -1;
}
