// Host-side helper of the quantizer step (a torch extension, plain C++: no kernels, no part of the C ABI).
//
// One PSQuantizer.record() + apply() over the ResNet-50 list touches 161 parameters from Python: read `param.grad`,
// collect and validate 76 device pointers for the multi-tensor launches, rebind 161 `.data` (ps_quantizer.py:63).  At
// ~0.1-0.15 us per attribute access that is ~60 us of a 150 us step whose kernels take 85 us (profiles/r03_host_breakdown.txt).
// The three loops below are those same loops in C++; gq_amd/quantizers.py falls back to its Python forms when this
// module is not built (same results, tests run both).
#include <torch/extension.h>

#include <cstdint>
#include <vector>

namespace {

// [p.grad for p in params]; an undefined gradient is an error exactly like `None.device` in the Python form
std::vector<at::Tensor> grads_of(const std::vector<at::Tensor> &params) {
    std::vector<at::Tensor> out;
    out.reserve(params.size());
    for (const at::Tensor &p : params) {
        const at::Tensor &g = p.grad();
        TORCH_CHECK(g.defined(), "gq_host.grads_of: a parameter has no gradient");
        out.push_back(g);
    }
    return out;
}

// The device pointers of `tensors` against the set of the last call (`last`, int64 CPU tensor of the same length,
// updated in place).  What the multi-tensor kernels assume about every tensor is checked on every call, also when the
// pointers did not move (a gradient replaced by a strided view or another dtype at the same address).
//   0: some tensor cannot be addressed (not contiguous, not float32, another device, misaligned)
//   1: valid, the same pointers as in `last`
//   2: valid, new pointers (now in `last`)
int64_t check_pointers(const std::vector<at::Tensor> &tensors, at::Tensor last, int64_t align, int64_t device_index) {
    TORCH_CHECK(last.device().is_cpu() && last.scalar_type() == at::kLong && last.is_contiguous() &&
                    last.numel() == (int64_t)tensors.size(),
                "gq_host.check_pointers: `last` must be a contiguous int64 CPU tensor with one entry per tensor");
    int64_t *lp = last.data_ptr<int64_t>();
    bool same = true;
    for (size_t i = 0; i < tensors.size(); ++i) {
        const at::Tensor &t = tensors[i];
        if (!t.defined() || t.scalar_type() != at::kFloat || !t.is_contiguous() || !t.device().is_cuda() ||
            t.device().index() != device_index)
            return 0;
        const int64_t p = (int64_t) reinterpret_cast<uintptr_t>(t.data_ptr());
        if (align > 1 && (p % align) != 0) return 0;
        if (lp[i] != p) {
            same = false;
            lp[i] = p;
        }
    }
    return same ? 1 : 2;
}

// for t, s in zip(targets, sources): t.data = s        (ps_quantizer.py:63 `param.grad.data = g`)
void rebind(const std::vector<at::Tensor> &targets, const std::vector<at::Tensor> &sources) {
    TORCH_CHECK(targets.size() == sources.size(), "gq_host.rebind: list lengths differ");
    for (size_t i = 0; i < targets.size(); ++i) {
        at::Tensor t = targets[i];
        t.set_data(sources[i]);
    }
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("grads_of", &grads_of, "[p.grad for p in params]");
    m.def("check_pointers", &check_pointers, "validate tensors for the multi-tensor launches and compare their pointers with the last set");
    m.def("rebind", &rebind, "t.data = s for every pair");
}
