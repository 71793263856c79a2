// Host-side marshalling of the three per-step calls (forward, backward range, Adam step) as a PyTorch C++ extension over the C ABI
// (include/wunet_hip.h): validates the ~100 - 400 tensors of a call, collects their addresses, allocates output / workspace from
// torch's caching allocator, takes torch's current HIP stream and calls libwunet_hip.so - what engine.py does through ctypes in
// ~0.2 ms of Python per step, here in ~20 us.  Optional: engine.py falls back to its ctypes path when this module is not built;
// the arithmetic is in libwunet_hip.so either way (BASELINE north_star: "exposed to Python ... via a PyTorch-ROCm C++/HIP extension";
// the reference's counterpart is torch's own dispatcher under model/unet_basic.py:77-100, trainer/trainer.py:36-38).
// Built by __graft_entry__.build() with g++ against torch's headers; no device code here.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <c10/core/DeviceGuard.h>

#include "wunet_hip.h"

namespace {

[[noreturn]] void fail_rc(int rc)
{
    throw std::runtime_error("wunet error " + std::to_string(rc) + ": " + wunet_last_error());
}

// engine.Engine._require for a whole list: float32 (int64 where said), contiguous, on the call's device
void require_all(const std::vector<at::Tensor>& ts, const char* what, const c10::Device& dev, at::ScalarType want = at::kFloat)
{
    for (size_t k = 0; k < ts.size(); ++k) {
        const at::Tensor& t = ts[k];
        if (t.scalar_type() != want) {
            std::ostringstream os;
            os << what << "[" << k << "]: expected " << (want == at::kFloat ? "float32" : "int64") << ", got " << t.scalar_type();
            throw std::runtime_error(os.str());
        }
        if (!t.is_contiguous()) throw std::runtime_error(std::string(what) + "[" + std::to_string(k) + "]: tensor must be contiguous");
        if (!t.is_cuda()) {
            std::ostringstream os;
            os << what << "[" << k << "]: the HIP path needs tensors on an MI355X (got device " << t.device() << "); there is no CPU fallback";
            throw std::runtime_error(os.str());
        }
        if (t.device() != dev) {
            std::ostringstream os;
            os << what << "[" << k << "] lives on " << t.device() << ", the call runs on " << dev << ": parameters and input live on different devices";
            throw std::runtime_error(os.str());
        }
    }
}

template <typename T>
std::vector<T*> pointers(const std::vector<at::Tensor>& ts)
{
    std::vector<T*> p(ts.size());
    for (size_t k = 0; k < ts.size(); ++k) p[k] = static_cast<T*>(ts[k].data_ptr());
    return p;
}

void* current_stream(const c10::Device& dev) { return (void*)c10::hip::getCurrentHIPStream(dev.index()).stream(); }

std::tuple<at::Tensor, at::Tensor> forward(int64_t handle, const at::Tensor& noisy, const std::vector<at::Tensor>& params,
                                           const std::vector<at::Tensor>& running, const std::vector<at::Tensor>& nbt, bool training,
                                           bool with_backward, const c10::optional<at::Tensor>& ws_reuse, bool packs_valid)
{
    wunet_ctx* h = reinterpret_cast<wunet_ctx*>(handle);
    const c10::Device dev = noisy.device();
    require_all({noisy}, "input", dev);
    require_all(params, "param", dev);
    require_all(running, "buffer", dev);
    require_all(nbt, "buffer", dev, at::kLong);
    c10::DeviceGuard guard(dev);
    const size_t nbytes = wunet_workspace_bytes(h, with_backward ? 1 : 0);
    // (eval mode: engine.Engine hands back the workspace of its previous eval forward on this ctx, packs_valid when the weights are unchanged)
    at::Tensor ws = (ws_reuse && (size_t)ws_reuse->numel() * 4 >= nbytes && ws_reuse->device() == dev) ? *ws_reuse
                                                                                                    : at::empty({(int64_t)(nbytes / 4)}, noisy.options());
    if (!(ws_reuse && ws.is_same(*ws_reuse))) packs_valid = false;
    at::Tensor out = at::empty_like(noisy);
    auto pp = pointers<const float>(params);
    auto rp = pointers<float>(running);
    auto np = pointers<long long>(nbt);
    const int rc = wunet_forward(h, noisy.data_ptr<float>(), pp.data(), rp.data(), np.data(), training ? 1 : 0,
                                 (with_backward ? WUNET_FWD_SAVE : 0) | (packs_valid && !training ? WUNET_FWD_PACKS_VALID : 0),
                                 ws.data_ptr(), out.data_ptr<float>(), current_stream(dev));
    if (rc) fail_rc(rc);
    return {out, ws};
}

// grads: ONE flat fp32 buffer + the element offset of every parameter (engine.FlatGrads)
void backward_range(int64_t handle, const at::Tensor& noisy, const std::vector<at::Tensor>& params, const at::Tensor& enhanced,
                    const at::Tensor& grad_enhanced, const at::Tensor& ws, const at::Tensor& flat, const std::vector<int64_t>& offsets,
                    int64_t layer_begin, int64_t layer_end, bool join)
{
    wunet_ctx* h = reinterpret_cast<wunet_ctx*>(handle);
    const c10::Device dev = noisy.device();
    require_all({grad_enhanced, flat}, "grad", dev);
    TORCH_CHECK(offsets.size() == params.size(), "one offset per parameter");
    c10::DeviceGuard guard(dev);
    auto pp = pointers<const float>(params);
    std::vector<float*> gp(offsets.size());
    float* const base = flat.data_ptr<float>();
    for (size_t k = 0; k < offsets.size(); ++k) gp[k] = base + offsets[k];
    const int rc = (join ? wunet_backward_range : wunet_backward_range_async)(h, noisy.data_ptr<float>(), pp.data(), enhanced.data_ptr<float>(),
                                                                             grad_enhanced.data_ptr<float>(), ws.data_ptr(), gp.data(),
                                                                             (int)layer_begin, (int)layer_end, current_stream(dev));
    if (rc) fail_rc(rc);
}

void adam_step(const std::vector<at::Tensor>& params, const std::vector<at::Tensor>& grads, const std::vector<at::Tensor>& exp_avg,
               const std::vector<at::Tensor>& exp_avg_sq, double lr, double beta1, double beta2, double eps, int64_t step, double grad_scale,
               const c10::optional<at::Tensor>& step_dev, const c10::optional<at::Tensor>& hyper_dev)
{
    TORCH_CHECK(!params.empty() && grads.size() == params.size() && exp_avg.size() == params.size() && exp_avg_sq.size() == params.size(),
                "adam_step: four lists of one length");
    const c10::Device dev = params[0].device();
    require_all(params, "adam param", dev);
    require_all(grads, "adam grad", dev);
    require_all(exp_avg, "adam exp_avg", dev);
    require_all(exp_avg_sq, "adam exp_avg_sq", dev);
    c10::DeviceGuard guard(dev);
    std::vector<size_t> numels(params.size());
    for (size_t k = 0; k < params.size(); ++k) numels[k] = (size_t)params[k].numel();
    auto pp = pointers<float>(params);
    auto gp = pointers<const float>(grads);
    auto mp = pointers<float>(exp_avg);
    auto vp = pointers<float>(exp_avg_sq);
    const int rc = wunet_adam_step((int)params.size(), pp.data(), gp.data(), mp.data(), vp.data(), numels.data(), lr, beta1, beta2, eps,
                                   (long long)step, grad_scale, step_dev ? (long long*)step_dev->data_ptr() : nullptr,
                                   hyper_dev ? hyper_dev->data_ptr<float>() : nullptr, current_stream(dev));
    if (rc) fail_rc(rc);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    // (the GIL is released for the whole call: the functions only read tensor metadata / pointers and allocate through ATen, both
    // thread-safe - stock nn.DataParallel enqueues its replicas from one thread each, trainer/base_trainer.py:26-27)
    m.def("forward", &forward, py::arg("handle"), py::arg("noisy"), py::arg("params"), py::arg("running"), py::arg("nbt"), py::arg("training"),
          py::arg("with_backward"), py::arg("ws_reuse") = py::none(), py::arg("packs_valid") = false, py::call_guard<py::gil_scoped_release>());
    m.def("backward_range", &backward_range, py::call_guard<py::gil_scoped_release>());
    m.def("adam_step", &adam_step, py::call_guard<py::gil_scoped_release>());
}
