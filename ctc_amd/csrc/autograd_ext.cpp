// Host-side fast path of the autograd Function (ctc_amd/functional.py): the SAME C-ABI calls as the Python Function,
// issued from a torch::autograd::Function in C++ so that an eager `CTCLoss.apply(...); loss.backward()` step
// (train.py:427,444) crosses Python twice instead of eight times.  No device code here and no HIP header: the
// entry points of libctc_amd.so arrive as raw function addresses (ctypes already resolved them), the stream as the
// integer torch hands out for the current stream, device memory through ATen's allocator.  Plumbing, not product:
// if this extension is missing the Python Function does the same work, slower on the host.
//
// Inputs are CANONICAL (the Python wrapper checks, and takes the general path otherwise): x float32 [T,B,C] on a HIP
// device with unit stride over C, targets on that device and contiguous ([B,S] int32/int64 or [B,S,C] float32),
// lengths int64 [B] on that device and contiguous.
#include <torch/extension.h>

#include <vector>

namespace {

typedef int (*noblank_fn)(const float *, int64_t, int64_t, const void *, int, const int64_t *, const int64_t *, int, int, int, int,
                          float, float, float *, float *, float *, void *, void *);
typedef int (*binary_fn)(const float *, int64_t, int64_t, const float *, const int64_t *, const int64_t *, int, int, int, int,
                         float, float, float *, float *, float *, void *, void *);
typedef int (*scale_fn)(float *, const float *, size_t, void *);

struct Abi {
    noblank_fn noblank = nullptr;
    binary_fn binary = nullptr;
    scale_fn scale = nullptr;
} g_abi;

// -> rc of the launch; fills loss / nll / grad
// (the workspace is the Python layer's: ctc_amd/functional.py::_workspace owns it and hands its address in)
int launch(int variant, const at::Tensor &x, const at::Tensor &targets, const at::Tensor &il, const at::Tensor &tl, double scale,
           at::Tensor &loss, at::Tensor &nll, at::Tensor *grad, int64_t stream, int64_t ws_ptr)
{
    const int T = (int)x.size(0), B = (int)x.size(1), C = (int)x.size(2), S = (int)targets.size(1);
    float *gp = grad ? grad->data_ptr<float>() : nullptr;
    void *st = reinterpret_cast<void *>(stream);
    void *wsp = reinterpret_cast<void *>(ws_ptr);
    if (variant == 0)
        return g_abi.noblank(x.data_ptr<float>(), x.stride(0), x.stride(1), targets.data_ptr(), targets.scalar_type() == at::kLong,
                             il.data_ptr<int64_t>(), tl.data_ptr<int64_t>(), T, B, C, S, (float)scale, (float)scale,
                             nll.data_ptr<float>(), loss.data_ptr<float>(), gp, wsp, st);
    return g_abi.binary(x.data_ptr<float>(), x.stride(0), x.stride(1), targets.data_ptr<float>(), il.data_ptr<int64_t>(),
                        tl.data_ptr<int64_t>(), T, B, C, S, (float)scale, (float)scale, nll.data_ptr<float>(),
                        loss.data_ptr<float>(), gp, wsp, st);
}

struct CtcFn : public torch::autograd::Function<CtcFn> {
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext *ctx, at::Tensor x, at::Tensor targets,
                                                  at::Tensor il, at::Tensor tl, int64_t variant, int64_t batch_total,
                                                  int64_t stream, int64_t ws_ptr, bool want)
    {
        const int64_t B = x.size(1);
        const double scale = 1.0 / (double)(batch_total > 0 ? batch_total : B);
        at::Tensor xd = x.detach();
        at::Tensor nll = at::empty({B}, xd.options());
        at::Tensor loss = at::empty({}, xd.options());
        at::Tensor grad;
        if (want) grad = at::empty({x.size(0), B, x.size(2)}, xd.options());
        const int rc = launch((int)variant, xd, targets, il, tl, scale, loss, nll, want ? &grad : nullptr, stream, ws_ptr);
        TORCH_CHECK(rc == 0, "ctc_amd: fused launch failed (", rc, ")");
        ctx->saved_data["variant"] = variant;
        ctx->saved_data["batch_total"] = batch_total;
        ctx->saved_data["stream"] = stream;
        ctx->saved_data["ws"] = ws_ptr;
        if (want) {
            ctx->saved_data["grad"] = grad;                    // handed to autograd exactly once
            ctx->save_for_backward({x, targets, il, tl});     // a second backward (retain_graph) recomputes
        }
        ctx->mark_non_differentiable({nll});
        return {loss, nll};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::variable_list gout)
    {
        // once differentiable, like the Python Function it stands in for (torch.autograd.function.once_differentiable):
        // the gradient below comes out of a raw kernel and carries no graph -- under create_graph=True it would silently be
        // treated as a constant
        // (the engine runs backward with grad mode ON exactly when create_graph=True)
        TORCH_CHECK(!at::GradMode::is_enabled(),
                    "ctc_amd: the CTC loss is differentiable once (create_graph=True / double backward is not supported)");
        at::Tensor grad;
        auto it = ctx->saved_data.find("grad");
        if (it != ctx->saved_data.end() && it->second.isTensor()) {
            grad = it->second.toTensor();
            ctx->saved_data["grad"] = c10::IValue();          // (None: the buffer now belongs to autograd)
        } else {
            auto saved = ctx->get_saved_variables();
            at::Tensor x = saved[0].detach();
            const int64_t stream = ctx->saved_data["stream"].toInt();   // (the recomputation uses the forward's stream and workspace)
            const int64_t B = x.size(1), bt = ctx->saved_data["batch_total"].toInt();
            at::Tensor nll = at::empty({B}, x.options()), loss = at::empty({}, x.options());
            grad = at::empty({x.size(0), B, x.size(2)}, x.options());
            const int rc = launch((int)ctx->saved_data["variant"].toInt(), x, saved[1], saved[2], saved[3],
                                  1.0 / (double)(bt > 0 ? bt : B), loss, nll, &grad, stream, ctx->saved_data["ws"].toInt());
            TORCH_CHECK(rc == 0, "ctc_amd: fused launch failed (", rc, ")");
        }
        at::Tensor g = gout[0];
        if (g.scalar_type() != at::kFloat || g.device() != grad.device() || !g.is_contiguous())
            g = g.to(grad.device(), at::kFloat).contiguous();
        // on the stream that is current NOW (the engine has made it wait for the forward's), as the Python Function does
        int64_t cur = 0;
        {
            pybind11::gil_scoped_acquire gil;
            // (leaked on purpose: a static pybind11 object would be destroyed after the interpreter is gone)
            static pybind11::object *raw = new pybind11::object(pybind11::module_::import("torch._C").attr("_cuda_getCurrentRawStream"));
            cur = (*raw)(grad.device().index()).cast<int64_t>();
        }
        const int rc = g_abi.scale(grad.data_ptr<float>(), g.data_ptr<float>(), (size_t)grad.numel(), reinterpret_cast<void *>(cur));
        TORCH_CHECK(rc == 0, "ctc_amd: scale_grad launch failed (", rc, ")");
        return {grad, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

void set_abi(int64_t noblank, int64_t binary, int64_t scale)
{
    g_abi.noblank = reinterpret_cast<noblank_fn>(noblank);
    g_abi.binary = reinterpret_cast<binary_fn>(binary);
    g_abi.scale = reinterpret_cast<scale_fn>(scale);
}

std::vector<at::Tensor> ctc_loss(at::Tensor x, at::Tensor targets, at::Tensor il, at::Tensor tl, int64_t variant,
                                 int64_t batch_total, int64_t stream, int64_t ws_ptr)
{
    TORCH_CHECK(g_abi.noblank != nullptr, "ctc_amd: set_abi() has not been called");
    const bool want = x.requires_grad() && at::GradMode::is_enabled();    // (inside forward() grad mode is already off)
    return CtcFn::apply(x, targets, il, tl, variant, batch_total, stream, ws_ptr, want);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("set_abi", &set_abi);
    m.def("ctc_loss", &ctc_loss);
}
