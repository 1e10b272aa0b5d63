// Host-side helper of gq_amd.quantizers.PSQuantizer (a CPython module, built by build.py with g++ against libtorch).
//
// With its launches replayed from HIP graphs, a quantizer step over a model's parameter list is bound by the HOST: per
// step the reference's protocol (quantizers/ps_quantizer.py:27-41 record, :43-63 apply) walks the parameters three
// times -- read every `param.grad`, look at its address / layout / dtype to decide whether a captured graph still
// describes it, and rebind `param.grad.data` to the mean.  161 tensors x ~0.5 us of interpreter work per visit is more
// than the kernels take.  The two walks below do the same visits in C++; nothing here computes on gradient data.
#include <torch/extension.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace py = pybind11;

// scan_grads(parameters) -> (grads, key, ok)
//   grads : the parameters' .grad tensors (the same Python objects `p.grad` returns; None where there is none)
//   key   : bytes, the grads' device addresses as int64 (dictionary key of the graph cache; 0 for a missing grad)
//   ok    : every grad is defined, float32, contiguous and on ONE device (what the captured launches assume)
static py::tuple scan_grads(const py::list &parameters) {
    const size_t n = parameters.size();
    py::list grads(n);
    std::string key(n * sizeof(int64_t), '\0');
    bool ok = true;
    c10::Device dev(c10::DeviceType::CPU);
    for (size_t i = 0; i < n; ++i) {
        if (!THPVariable_Check(parameters[i].ptr())) throw py::type_error("scan_grads: the list holds something that is not a tensor");
        const at::Tensor &p = THPVariable_Unpack(parameters[i].ptr());
        const at::Tensor &g = p.grad();
        int64_t ptr = 0;
        if (g.defined()) {
            ptr = reinterpret_cast<int64_t>(g.data_ptr());
            ok = ok && g.scalar_type() == at::kFloat && g.is_contiguous();
            if (i == 0)
                dev = g.device();
            else
                ok = ok && g.device() == dev;
            grads[i] = py::reinterpret_steal<py::object>(THPVariable_Wrap(g));
        } else {
            ok = false;
            grads[i] = py::none();
        }
        std::memcpy(&key[i * sizeof(int64_t)], &ptr, sizeof(int64_t));
    }
    return py::make_tuple(grads, py::bytes(key), ok);
}

// scan_key(parameters) -> (key, ok, device index): scan_grads without the list of Python objects -- what a record() that replays a
// whole-step graph needs to know (161 object look-ups less per step on a path the host bounds); device index -1: no CUDA gradient.
static py::tuple scan_key(const py::list &parameters) {
    const size_t n = parameters.size();
    std::string key(n * sizeof(int64_t), '\0');
    bool ok = n > 0;
    c10::Device dev(c10::DeviceType::CPU);
    for (size_t i = 0; i < n; ++i) {
        if (!THPVariable_Check(parameters[i].ptr())) throw py::type_error("scan_key: the list holds something that is not a tensor");
        const at::Tensor &g = THPVariable_Unpack(parameters[i].ptr()).grad();
        int64_t ptr = 0;
        if (g.defined()) {
            ptr = reinterpret_cast<int64_t>(g.data_ptr());
            ok = ok && g.scalar_type() == at::kFloat && g.is_contiguous();
            if (i == 0)
                dev = g.device();
            else
                ok = ok && g.device() == dev;
        } else {
            ok = false;
        }
        std::memcpy(&key[i * sizeof(int64_t)], &ptr, sizeof(int64_t));
    }
    return py::make_tuple(py::bytes(key), ok, dev.is_cuda() ? (int)dev.index() : -1);
}

// `o.data = v` for one tensor.  at::Tensor::set_data re-derives everything a tensor carries from v (sizes, strides, key set, the
// autograd meta's accumulator, ...): ~0.12 us, and a quantizer step does it 161 times in apply() (and the caller's backward, or
// bench.py's feeder, as many times again) -- the largest single item of a replayed step's host time.  When v has the layout o
// already has (same sizes, strides, dtype, device, offset 0 on both or not: what `.data =` would copy anyway), the ONLY thing
// that changes is which storage the tensor looks at: swap that and leave the rest.  Anything else (another shape, a tensor whose
// metadata may not change -- the result of .detach() --, a subclass, sparse / quantized layouts) takes set_data itself.
static inline void rebind(const at::Tensor &o, const at::Tensor &v) {
    c10::TensorImpl *oi = o.unsafeGetTensorImpl(), *vi = v.unsafeGetTensorImpl();
    if (oi != vi && oi->allow_tensor_metadata_change() && oi->key_set() == vi->key_set() && oi->dtype() == vi->dtype() &&
        oi->layout() == c10::kStrided && vi->layout() == c10::kStrided && oi->has_storage() && vi->has_storage() &&
        oi->device() == vi->device() && oi->sizes().equals(vi->sizes()) && oi->strides().equals(vi->strides()) &&
        !oi->is_python_dispatch() && !vi->is_python_dispatch() && !oi->has_named_tensor_meta() && !vi->has_named_tensor_meta()) {
        oi->set_storage_keep_dtype(vi->storage());
        oi->set_storage_offset(vi->storage_offset());
        return;
    }
    o.set_data(v);
}

// set_data(objects, values): objects[i].data = values[i]  (ps_quantizer.py:63 for every parameter)
static void set_data(const py::list &objects, const py::list &values) {
    const size_t n = objects.size();
    if (values.size() != n) throw std::invalid_argument("set_data: the two lists differ in length");
    for (size_t i = 0; i < n; ++i) {
        if (!THPVariable_Check(objects[i].ptr()) || !THPVariable_Check(values[i].ptr()))
            throw py::type_error("set_data: the lists hold something that is not a tensor");
        const at::Tensor &o = THPVariable_Unpack(objects[i].ptr());
        const at::Tensor &v = THPVariable_Unpack(values[i].ptr());
        rebind(o, v);
    }
}

// set_grad_data(parameters, values): parameters[i].grad.data = values[i], with `param.grad` evaluated NOW, as
// ps_quantizer.py:63 does -- a caller that replaced a parameter's .grad object between the last record() and apply()
// gets the mean in the object it holds now.  (p.grad() in C++ builds no Python object: this costs what set_data on the
// remembered objects did.)  A parameter without a gradient raises AttributeError, as `None.data = g` does.
static void set_grad_data(const py::list &parameters, const py::list &values) {
    const size_t n = parameters.size();
    if (values.size() != n) throw std::invalid_argument("set_grad_data: the two lists differ in length");
    for (size_t i = 0; i < n; ++i) {
        if (!THPVariable_Check(parameters[i].ptr()) || !THPVariable_Check(values[i].ptr()))
            throw py::type_error("set_grad_data: the lists hold something that is not a tensor");
        const at::Tensor &p = THPVariable_Unpack(parameters[i].ptr());
        const at::Tensor &g = p.grad();
        if (!g.defined()) throw py::attribute_error("'NoneType' object has no attribute 'data' (parameter " + std::to_string(i) + " has no .grad)");
        rebind(g, THPVariable_Unpack(values[i].ptr()));
    }
}

PYBIND11_MODULE(_gq_host, m) {
    m.def("set_grad_data", &set_grad_data, "parameters[i].grad.data = values[i], .grad evaluated at the call");
    m.def("scan_grads", &scan_grads, "The .grad tensors of a parameter list, their addresses as a bytes key, and whether all are plain f32");
    m.def("set_data", &set_data, "objects[i].data = values[i]");
    m.def("scan_key", &scan_key, "The parameters' gradient addresses as a bytes key, whether all are plain f32 on one device, that device's index");
}
