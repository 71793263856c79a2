// Instantiation unit: conv_h3d_kernel<TAPS, M_REP, NSEG, BF> (fp16-split conv / data gradient, DMA-staged and pipelined).
#include "wunet_h3d.h"
#include "wunet_launch.h"

#define WUNET_XCASE(T, M, S)                                                                               \
    if (taps == T && mrep == M && nseg == S) {                                                             \
        if (!WUNET_H3D_HAS_TAIL(M, S) && a.NFS != a.NS) return -3;      /* pack with a K tail, kernel without */ \
        if (bf) {                                                                                          \
            if (WUNET_ALLOW_BIG_LDS((conv_h3d_kernel<T, M, S, true>), smem) != 0) return -2;               \
            WUNET_LAUNCH((conv_h3d_kernel<T, M, S, true>), grid, dim3(WUNET_THREADS), smem, st, a);        \
        } else {                                                                                           \
            if (WUNET_ALLOW_BIG_LDS((conv_h3d_kernel<T, M, S>), smem) != 0) return -2;                     \
            WUNET_LAUNCH((conv_h3d_kernel<T, M, S>), grid, dim3(WUNET_THREADS), smem, st, a);              \
        }                                                                                                  \
        return 0;                                                                                          \
    }

// EVOP: the eval-mode encoder form that also writes the next layer's operand (15 taps, whole-row tiles, two planes)
#define WUNET_XCASE_EVOP(M)                                                                                \
    if (evop && taps == 15 && mrep == M && nseg == 1 && !bf) {                                             \
        if (!WUNET_H3D_HAS_TAIL(M, 1) && a.NFS != a.NS) return -3;                                         \
        if (WUNET_ALLOW_BIG_LDS((conv_h3d_kernel<15, M, 1, false, true>), smem) != 0) return -2;           \
        WUNET_LAUNCH((conv_h3d_kernel<15, M, 1, false, true>), grid, dim3(WUNET_THREADS), smem, st, a);    \
        return 0;                                                                                          \
    }

// BSUM: the training data gradient that also takes the producers' BatchNorm-backward sums (whole-row tiles, two planes): 5 taps = a decoder
// layer (rows through the upsample + skip rows), 15 taps = an encoder-side layer (decimated rows)
#define WUNET_XCASE_BSUM(T, M, K)                                                                          \
    if (bsum == K && taps == T && mrep == M && nseg == 1 && !bf && !evop) {                                \
        if (!WUNET_H3D_HAS_TAIL(M, 1) && a.NFS != a.NS) return -3;                                         \
        if (WUNET_ALLOW_BIG_LDS((conv_h3d_kernel<T, M, 1, false, false, K>), smem) != 0) return -2;        \
        WUNET_LAUNCH((conv_h3d_kernel<T, M, 1, false, false, K>), grid, dim3(WUNET_THREADS), smem, st, a); \
        return 0;                                                                                          \
    }

int wunet_launch_conv_h3d(const ConvH3Args& a, int taps, int mrep, int nseg, dim3 grid, size_t smem, hipStream_t st, bool bf, bool evop, int bsum)
{
    WUNET_XCASE_BSUM(5, 2, 1) WUNET_XCASE_BSUM(5, 3, 1) WUNET_XCASE_BSUM(5, 4, 1)
    WUNET_XCASE_BSUM(15, 2, 2) WUNET_XCASE_BSUM(15, 3, 2) WUNET_XCASE_BSUM(15, 4, 2)
    WUNET_XCASE_BSUM(5, 2, 3) WUNET_XCASE_BSUM(5, 3, 3) WUNET_XCASE_BSUM(5, 4, 3)           // UPT: the upsampled rows stored at the producer's resolution
    if (bsum) return -5;
    WUNET_XCASE_EVOP(2) WUNET_XCASE_EVOP(3) WUNET_XCASE_EVOP(4)
    if (evop) return -4;
    WUNET_XCASE(15, 2, 1) WUNET_XCASE(15, 3, 1) WUNET_XCASE(15, 4, 1)
    WUNET_XCASE(5, 2, 1) WUNET_XCASE(5, 3, 1) WUNET_XCASE(5, 4, 1)
    WUNET_XCASE(15, 2, 2) WUNET_XCASE(15, 3, 2) WUNET_XCASE(5, 2, 2) WUNET_XCASE(5, 3, 2)
    WUNET_XCASE(15, 2, 4) WUNET_XCASE(15, 3, 4) WUNET_XCASE(5, 2, 4) WUNET_XCASE(5, 3, 4)
    WUNET_XCASE(15, 2, 8) WUNET_XCASE(15, 3, 8) WUNET_XCASE(5, 2, 8) WUNET_XCASE(5, 3, 8)
    WUNET_XCASE(15, 2, 16) WUNET_XCASE(15, 3, 16) WUNET_XCASE(5, 2, 16) WUNET_XCASE(5, 3, 16)
    return -1;
}

