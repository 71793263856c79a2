/*
 * C ABI of the MI355X-native Wave-U-Net hot path (libwunet_hip.so, gfx950).
 *
 * The reference (haoxiangsnr/Wave-U-Net-for-Speech-Enhancement) is pure Python: its "FFI" for
 * this path is torch's dispatcher.  These entry points are what a maintainer binds instead
 * (ctypes stub in INTEGRATION.md); each cites the reference interface it replaces.
 * All pointers are DEVICE pointers (hipMalloc'ed / torch caching allocator) unless noted, fp32,
 * layout (batch, channel, sample) contiguous.  `stream` is a hipStream_t (NULL = default stream).
 * Every function returns 0 on success and a negative code on failure; wunet_last_error() gives text.
 * Nothing here synchronises the stream: calls only enqueue kernels.
 *
 * Parameter order ("canonical order", == nn.Module.parameters() order of the reference Model,
 * model/unet_basic.py:33-75): for each conv layer in forward order
 * [encoder.0 .. encoder.n-1, middle, decoder.0 .. decoder.n-1]:
 *   conv.weight [Cout,Cin,K], conv.bias [Cout], bn.weight [Cout], bn.bias [Cout];
 * then out.0.weight [1,ci+1,1], out.0.bias [1].   -> 4*(2n+1)+2 pointers.
 * Running statistics: 2*(2n+1) pointers (running_mean, running_var per layer);
 * num_batches_tracked: (2n+1) int64 device pointers.
 */
#ifndef WUNET_HIP_H
#define WUNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wunet_ctx wunet_ctx;

#define WUNET_OK 0
#define WUNET_E_ARG (-1)        /* bad argument / unsupported shape */
#define WUNET_E_RUNTIME (-2)    /* HIP runtime error */

#define WUNET_LOSS_MSE 0        /* model/loss.py:3-4  torch.nn.MSELoss()      */
#define WUNET_LOSS_L1 1         /* model/loss.py:6-7  torch.nn.L1Loss()       */
#define WUNET_LOSS_SMOOTH_L1 2  /* torch.nn.SmoothL1Loss(beta=1), SURVEY.md §0 */

const char* wunet_last_error(void);

/* Replaces Model.__init__ shape bookkeeping (model/unet_basic.py:33-75) for one (batch, length).
 * length >= 4 and divisible by 2^n_layers, as model/unet_basic.py:86,93 requires (levels of 1-2 samples take a scalar path).  The
 * kernels index rows with shifts and masks: a length that is not a power of two (m * 2^k) is carried in rows padded to the next
 * power of two inside the workspace (zeros where a conv reads the padding, none of it in the BatchNorm statistics, no gradient
 * for it, the upsample at the coordinates of the lengths that exist); the caller's tensors keep their own [B][1][length] layout.
 * Host-side object, no device memory. */
int wunet_create(int n_layers, int channels_interval, int batch, int length, wunet_ctx** out);
void wunet_destroy(wunet_ctx* ctx);

/* GEMM arithmetic of the levels >= 16 samples.  enable = 1: forward convs, data gradients and weight
 * gradients run as fp16-split GEMMs - every fp32 operand x is carried as hi + lo in fp16 (22 significant bits; weights,
 * activations and gradients each with a power-of-two scale derived on the device, so the result does not depend on the
 * magnitude of the checkpoint), a product is three v_mfma_f32_16x16x32_f16 passes with fp32 accumulation -
 * wherever the position grid fills the chip (levels >= 256 samples; 128 .. 16 samples at >= 1024 positions per level,
 * with split-K); 2: wherever the kernels can run (>= 16 samples; small test shapes); 0: fp32 MFMA
 * (v_mfma_f32_16x16x4_f32) everywhere.  Accuracy of the split path is at the fp32 noise floor, but the
 * arithmetic is not bit-identical to the fp32 path.  3 / 4: the layer sets of 1 / 2 in the bf16 mode of the same kernels -
 * operands stored as one bf16 word, one v_mfma_f32_16x16x32_bf16 pass, fp32 accumulation (BASELINE.json configs[4]; accuracy
 * is that of bf16: ~1e-2, tests/test_bf16_mode.py).  A new ctx starts with 0; the Python Engine turns 1 on unless
 * WUNET_H3 says otherwise.  Changes the workspace size: call before wunet_workspace_bytes.  No reference counterpart. */
int wunet_set_h3(wunet_ctx* ctx, int enable);

/* Bytes of device workspace the caller must provide to wunet_forward (with_backward=0: inference;
 * with_backward=1: also holds everything wunet_backward needs). */
size_t wunet_workspace_bytes(const wunet_ctx* ctx, int with_backward);

/* Replaces Model.forward (model/unet_basic.py:77-100).
 * training != 0: BatchNorm uses batch statistics and updates running_mean/var (momentum 0.1,
 * unbiased variance) and num_batches_tracked in place; otherwise running statistics are used.
 * save_for_backward: bit 0 (WUNET_FWD_SAVE, needs a with_backward=1 workspace): additionally keeps what wunet_backward
 * needs - the raw conv outputs, BN statistics and each conv's activated input.
 * Bit 1 (WUNET_FWD_PACKS_VALID, eval mode only): the caller asserts that `workspace` is the one the previous eval-mode wunet_forward of
 * this ctx ran on and that no conv weight has changed since - the weight packs and weight scales it holds are reused and the three
 * pack launches are skipped (enhancement.py:57-69 runs the same weights over every chunk of every file; engine.Engine keys this on the
 * parameters' addresses and autograd version counters). */
#define WUNET_FWD_SAVE 1
#define WUNET_FWD_PACKS_VALID 2
int wunet_forward(wunet_ctx* ctx, const float* noisy, const float* const* params,
                  float* const* running, long long* const* num_batches_tracked, int training,
                  int save_for_backward, void* workspace, float* enhanced, void* stream);

/* Replaces autograd's backward of Model.forward (trainer/trainer.py:37 loss.backward()).
 * Must follow a training-mode wunet_forward on the same ctx/workspace/params/noisy.
 * grad_enhanced: dL/d(enhanced) [B,1,T].  grads: same shapes/order as params; every tensor is
 * overwritten (conv biases feeding training-mode BatchNorm get an exact 0).  No input gradient. */
int wunet_backward(wunet_ctx* ctx, const float* noisy, const float* const* params,
                   const float* enhanced, const float* grad_enhanced, void* workspace,
                   float* const* grads, void* stream);
/* Same as wunet_backward but only runs conv layers [layer_begin, layer_end) in backward order
 * (layer index in forward order; the output head belongs to layer 2n).  Lets the caller overlap the
 * RCCL all-reduce of finished gradient buckets with the remaining layers. */
int wunet_backward_range(wunet_ctx* ctx, const float* noisy, const float* const* params,
                         const float* enhanced, const float* grad_enhanced, void* workspace,
                         float* const* grads, int layer_begin, int layer_end, void* stream);
/* wunet_backward_range without the final "caller's stream waits for the weight-gradient stream": the next range's
 * data-gradient chain is not held up at the bucket boundary.  The weight gradients of the range are complete for a stream
 * after wunet_backward_join(ctx, that_stream) - e.g. the stream the RCCL all-reduce of the bucket is enqueued from.  The range
 * with layer_begin == 0 (the end of the backward) always joins `stream`.  The side stream is per (ctx, device): a ctx may be
 * used from several devices / threads at once as long as each call runs with its device current. */
int wunet_backward_range_async(wunet_ctx* ctx, const float* noisy, const float* const* params,
                               const float* enhanced, const float* grad_enhanced, void* workspace,
                               float* const* grads, int layer_begin, int layer_end, void* stream);
int wunet_backward_join(wunet_ctx* ctx, void* stream);

/* Replaces loss_function(clean, enhanced) (trainer/trainer.py:36; model/loss.py:3-7), mean reduction.
 * scratch: >= wunet_loss_scratch_bytes() device bytes.  loss_out: device float. */
size_t wunet_loss_scratch_bytes(void);
int wunet_loss_forward(int kind, const float* clean, const float* enhanced, size_t n,
                       float* loss_out, void* scratch, void* stream);
/* grad_enhanced[i] = grad_loss[0] * d loss / d enhanced[i]  (grad_loss: device float, usually 1). */
int wunet_loss_backward(int kind, const float* clean, const float* enhanced, const float* grad_loss,
                        size_t n, float* grad_enhanced, void* stream);

/* replaces optimizer.step() of torch.optim.Adam(params, lr, betas) as the reference builds it
 * (train.py:31-35; eps, weight_decay=0, amsgrad=False fixed by the reference's call) for n_tensors tensors in one
 * or two launches.  step is the 1-based step count AFTER the increment (torch's state["step"]).  numels: host array.
 * grad_scale multiplies every gradient first (1/world_size of the data-parallel average, trainer/base_trainer.py:26-27, folded
 * into the step).  step_dev != NULL: the step count lives in device memory (int64), is incremented by the call and the bias
 * corrections are computed on the device into hyper_dev (2 floats) - `step` is ignored; this is what makes the whole training
 * step capturable in a hipGraph. */
int wunet_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const size_t* numels, double lr, double beta1, double beta2, double eps,
                    long long step, double grad_scale, long long* step_dev, float* hyper_dev, void* stream);

/* Introspection for tests / profiling: float offset of layer i's raw conv output inside the
 * workspace, its channel count and length. */
int wunet_layer_info(const wunet_ctx* ctx, int layer, size_t* z_offset_floats, int* channels, int* length);
int wunet_num_conv_layers(const wunet_ctx* ctx);

/* Optional per-launch timing of the MFMA kernels with HIP events recorded on the launch stream
 * (bench.py's roofline leg; no reference counterpart).  collect() synchronises the device, writes
 * "kernel name\tlaunches\ttotal_ms\ttotal_algorithmic_flops\ttotal_algorithmic_bytes\n" per kernel
 * into buf and clears the records. */
int wunet_profile_enable(int on);
long long wunet_profile_collect(char* buf, size_t cap);
/* Measurement hook (environment WUNET_STAMP=1, otherwise no launches): one-thread kernels write the device's 100 MHz wall clock beside the
 * backward's launches - slot 2 i in front of layer i's data gradient on the caller's stream, slot 2 i + 1 in front of its weight gradient
 * on the library's side stream, slots 126 / 127 where the two chains join.  They are captured into a step graph like any launch, so a graph
 * REPLAY can be asked when its two chains ran without a tracer attached.  Copies up to 128 values of the last backward to `out`. */
int wunet_debug_stamps(unsigned long long* out, int n);

/* ---- data-parallel exchange: the flat fp32 gradient buffer summed over the GPUs of the job by RCCL.
 * Replaces what `torch.nn.DataParallel(model, device_ids=...)` does implicitly per step - replicate / scatter / gather and the
 * reduce-add of the replicas' gradients onto device 0 (/root/reference/trainer/base_trainer.py:26-27) - with ONE process per GPU
 * and an all-reduce (sum) over xGMI; the caller scales by 1/world (or leaves it to wunet_adam_step's grad_scale).
 * wunet_comm_unique_id: rank 0 draws the 128-byte RCCL id; the caller carries it to the other ranks (any side channel: a file, a
 * socket, torch.distributed's store).  wunet_comm_create: collective over all `world` ranks, on the calling thread's current device.
 * wunet_comm_allreduce_sum: in place over buf[0 .. count), enqueued on `stream` like a kernel: it orders with the backward's kernels
 * on that stream and can be captured into the step's hipGraph.  World size 1 needs no RCCL (the sum over one rank is a no-op). */
#define WUNET_COMM_ID_BYTES 128
typedef struct wunet_comm wunet_comm;
int wunet_comm_unique_id(unsigned char* id /* [WUNET_COMM_ID_BYTES] */);
int wunet_comm_create(const unsigned char* id, int world, int rank, wunet_comm** out);
int wunet_comm_allreduce_sum(wunet_comm* comm, float* buf, size_t count, void* stream);
int wunet_comm_world(const wunet_comm* comm);
void wunet_comm_destroy(wunet_comm* comm);

/* SURVEY.md section 8 (f4), the data input path.  Replaces, for a whole batch and on the device, the reference's per-item aligned
 * random crop - `sample_fixed_length_data_aligned(mixture, clean, sample_length)`, /root/reference/util/utils.py:101-113, called from
 * `Dataset.__getitem__`, dataset/waveform_dataset.py:56-67 - and the collate of train.py:15-21: window b of BOTH outputs is
 * samples starts[b] .. starts[b] + length - 1 of the two flat float32 arrays (`total` samples each, resident in HBM; starts = device
 * int64, drawn by the caller with the reference's distribution: uniform over [item_start, item_start + item_length - length]).
 * mixture / clean receive [batch][1][length] float32.  Starts outside [0, total - length] are clamped (never an out-of-range read). */
int wunet_crop_windows(const float* mixture_flat, const float* clean_flat, const long long* starts, long long total,
                       int batch, int length, float* mixture, float* clean, void* stream);

/* Measurement aid (tools/conv_trace.py; no reference counterpart): a device buffer of gridDim.x * 64 uint64 that the DMA-staged
 * conv / data-gradient kernel (conv_h3d_kernel) fills with shader-clock stamps of each block's phases (entry; per K stage: tile
 * landed, W buffer released, MFMAs issued; epilogue done), or NULL (default) to turn the stamps off.  Process-wide. */
void wunet_debug_set_conv_trace(void* dev_buffer);

/* Single-op entry points (parity tests of the MFMA kernels against F.conv1d semantics,
 * nn.Conv1d stride 1, padding K/2, K in {5, 15}).  They allocate scratch and synchronise. */
int wunet_op_conv1d(const float* x, const float* w, const float* bias, float* z,
                    int B, int Cin, int Cout, int L, int K, void* stream);
int wunet_op_conv1d_dgrad(const float* gz, const float* w, float* dx,
                          int B, int Cin, int Cout, int L, int K, void* stream);
int wunet_op_conv1d_wgrad(const float* gz, const float* x, float* dw,
                          int B, int Cin, int Cout, int L, int K, void* stream);

/* The same three ops on the fp16-split kernels (conv_h3_kernel / wgrad_h3d_kernel / wgrad_h3_kernel of csrc/wunet_h3.h) with
 * the planner's tiling for the geometry and device-derived power-of-two operand scales: L >= 16, B*L >= 256, and Cin >= 16
 * for the two gradients (the network planner's own conditions).  They allocate scratch and synchronise. */
int wunet_op_conv1d_split(const float* x, const float* w, const float* bias, float* z,
                          int B, int Cin, int Cout, int L, int K, void* stream);
int wunet_op_conv1d_dgrad_split(const float* gz, const float* w, float* dx,
                                int B, int Cin, int Cout, int L, int K, void* stream);
int wunet_op_conv1d_wgrad_split(const float* gz, const float* x, float* dw,
                                int B, int Cin, int Cout, int L, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif
