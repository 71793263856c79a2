// Launch plumbing shared by the host API and the per-(taps, mode) instantiation units.
#pragma once
#include "wunet_kernels.h"

#ifdef WUNET_EMU
#define WUNET_LAUNCH(kern, grid, block, smem, stream, ...) emu::launch(grid, block, smem, [=]() { kern(__VA_ARGS__); })
#define WUNET_ALLOW_BIG_LDS(kern, smem) 0
#else
#define WUNET_LAUNCH(kern, grid, block, smem, stream, ...) hipLaunchKernelGGL(kern, grid, block, smem, stream, __VA_ARGS__)
// gfx950 has 160 KiB of LDS per CU; more than 64 KiB of dynamic LDS must be requested per kernel
// (granted once per kernel and device, remembered: hipFuncSetAttribute on every launch costs host time)
#define WUNET_ALLOW_BIG_LDS(kern, smem)                                                                    \
    ([&]() -> int {                                                                                        \
        static size_t granted[64];                                                                         \
        if ((size_t)(smem) <= 64 * 1024) return 0;                                                         \
        int dev = 0;                                                                                       \
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, granted[0] = 0;             \
        if ((size_t)(smem) <= granted[dev]) return 0;                                                      \
        const int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&kern),                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(smem));   \
        if (e == 0) granted[dev] = (size_t)(smem);                                                         \
        return e;                                                                                          \
    }())
#endif

// One translation unit per tap count keeps hipcc builds parallel.  Return 0 when a kernel
// for (mrep, nrep) exists and was enqueued, -1 otherwise.
#define WUNET_DECL_CONV(T) int wunet_launch_conv_##T(const ConvArgs& a, int mrep, int nrep, dim3 grid, size_t smem, hipStream_t st)
#define WUNET_DECL_WGRAD(T) int wunet_launch_wgrad_##T(const WgradArgs& a, int mrep, int nw, int xit, int wsplit, dim3 grid, size_t smem, hipStream_t st)
WUNET_DECL_CONV(15);
WUNET_DECL_CONV(5);
WUNET_DECL_WGRAD(15);
WUNET_DECL_WGRAD(5);

struct ConvH3Args;
struct ConvH3uArgs;
struct WgradH3Args;
struct WgradH3dArgs;
int wunet_launch_wgrad_h3d(const WgradH3dArgs& a, int taps, int mrep, bool db, dim3 grid, size_t smem, hipStream_t st, bool bf, int tp);
int wunet_launch_conv_h3d(const ConvH3Args& a, int taps, int mrep, int nseg, dim3 grid, size_t smem, hipStream_t st, bool bf, bool evop = false, int bsum = 0);
int wunet_launch_conv_h3u(const ConvH3uArgs& a, int mrep, dim3 grid, size_t smem, hipStream_t st);
int wunet_launch_wgrad_h3(const WgradH3Args& a, int taps, int mrep, int nseg, int tp, dim3 grid, size_t smem, hipStream_t st, bool bf);
