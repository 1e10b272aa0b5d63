// Library-level entry points and shared host helpers.
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "gq_common.hpp"

namespace gq {

char *last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-dispatch timing (profile_slot of the encode entry points / gq_profile_read) -----------------------
static hipEvent_t g_prof_events[GQ_PROFILE_SLOTS][2];
static bool g_prof_created[GQ_PROFILE_SLOTS];

bool profile_events(int slot, hipEvent_t *start, hipEvent_t *stop) {
    if (slot < 0 || slot >= GQ_PROFILE_SLOTS) return false;
    if (!g_prof_created[slot]) {
        if (hipEventCreate(&g_prof_events[slot][0]) != hipSuccess || hipEventCreate(&g_prof_events[slot][1]) != hipSuccess)
            return false;
        g_prof_created[slot] = true;
    }
    *start = g_prof_events[slot][0];
    *stop = g_prof_events[slot][1];
    return true;
}

int cu_count() {
    // compute units per device, looked up once per device (a plain table: a racing first use writes the same value)
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

__global__ void axpy_inplace_kernel(float *__restrict__ y, const float *__restrict__ x, float a, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        // ps_quantizer.py:35  grad.add_(scale * error): the product is rounded, then added
        float t = a * x[i];
        y[i] = y[i] + t;
    }
}

__global__ void sub_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ o,
                           int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) o[i] = a[i] - b[i];
}

static inline int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    int64_t cap = (int64_t)cu_count() * 8;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace gq

GQ_API int gq_abi_version(void) { return 5; }

GQ_API const char *gq_last_error(void) { return gq::last_error_buf(); }

GQ_API int gq_device_info(int device, int *cu_count, char *arch, size_t arch_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return gq::fail(GQ_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (arch && arch_len) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return GQ_OK;
}

GQ_API int gq_axpy_inplace(float *grad, const float *err, float scale, int64_t n, void *stream) {
    if (n < 0 || (n > 0 && (!grad || !err))) return gq::fail(GQ_ERR_INVALID_ARG, "gq_axpy_inplace: bad arguments");
    if (n == 0) return GQ_OK;
    hipLaunchKernelGGL(gq::axpy_inplace_kernel, dim3(gq::grid_for(n, 256)), dim3(256), 0, gq::as_stream(stream), grad,
                       err, scale, n);
    GQ_CHECK_LAUNCH("gq_axpy_inplace");
    return GQ_OK;
}

GQ_API int gq_sub(const float *grad, const float *decoded, float *err, int64_t n, void *stream) {
    if (n < 0 || (n > 0 && (!grad || !decoded || !err))) return gq::fail(GQ_ERR_INVALID_ARG, "gq_sub: bad arguments");
    if (n == 0) return GQ_OK;
    hipLaunchKernelGGL(gq::sub_kernel, dim3(gq::grid_for(n, 256)), dim3(256), 0, gq::as_stream(stream), grad, decoded,
                       err, n);
    GQ_CHECK_LAUNCH("gq_sub");
    return GQ_OK;
}

// ---- mean of R rows (the aggregate of identity-compressed tensors) -------------------------------------------
namespace gq {
// (reset_dst <- reset_src, reset_words 64-bit words: the accumulators the NEXT step's kernels fold into go back to their empty
// state in the step's last launch -- see gq_mean_rows in gq_hsq.h)
__device__ __forceinline__ void copy_reset_words(uint64_t *__restrict__ dst, const uint64_t *__restrict__ src, int words) {
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void mean_rows_kernel(const uint8_t *__restrict__ rows, int64_t row_stride_bytes, int R,
                                                        int64_t n, float *__restrict__ out, uint64_t *rng_state, int rng_pairs,
                                                        uint64_t *reset_dst, const uint64_t *reset_src, int reset_words) {
    bump_rng_counter(rng_state, rng_pairs);
    copy_reset_words(reset_dst, reset_src, reset_words);
    const MeanDiv md = mean_div_of(R, true);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float acc = reinterpret_cast<const float *>(rows)[i];
        for (int r = 1; r < R; ++r) acc = acc + reinterpret_cast<const float *>(rows + (int64_t)r * row_stride_bytes)[i];
        out[i] = mean_div(acc, md);   // (+0 + row 0 + row 1 + ...) / R, rows ascending, a true division: torch's CPU mean
    }
}
__global__ void rng_step_kernel(uint64_t *rng_state, int rng_pairs, uint64_t *reset_dst, const uint64_t *reset_src, int reset_words) {
    bump_rng_counter(rng_state, rng_pairs);
    copy_reset_words(reset_dst, reset_src, reset_words);
}
}  // namespace gq

GQ_API int gq_mean_rows(const void *rows, int64_t row_stride_bytes, int R, int64_t n, float *out, uint64_t *rng_state,
                        int rng_pairs, uint64_t *reset_dst, const uint64_t *reset_src, int reset_words, void *stream) {
    if (R < 1 || n < 0 || (n > 0 && (!rows || !out)) || (row_stride_bytes & 3) != 0 || (rng_state && (rng_pairs < 1 || rng_pairs > 256)) ||
        reset_words < 0 || (reset_words > 0 && (!reset_dst || !reset_src)))
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_mean_rows: bad arguments");
    if (n == 0 && !rng_state && reset_words == 0) return GQ_OK;
    if (n == 0)
        hipLaunchKernelGGL(gq::rng_step_kernel, dim3(1), dim3(256), 0, gq::as_stream(stream), rng_state, rng_state ? rng_pairs : 0,
                           reset_dst, reset_src, reset_words);
    else
        hipLaunchKernelGGL(gq::mean_rows_kernel, dim3(gq::grid_for(n, 256)), dim3(256), 0, gq::as_stream(stream),
                           static_cast<const uint8_t *>(rows), row_stride_bytes, R, n, out, rng_state, rng_pairs, reset_dst,
                           reset_src, reset_words);
    GQ_CHECK_LAUNCH("gq_mean_rows");
    return GQ_OK;
}

GQ_API int gq_profile_read(int slot, float *kernel_ms) {
    if (slot < 0 || slot >= GQ_PROFILE_SLOTS || !kernel_ms || !gq::g_prof_created[slot])
        return gq::fail(GQ_ERR_INVALID_ARG, "gq_profile_read: slot %d was never armed", slot);
    if (hipEventSynchronize(gq::g_prof_events[slot][1]) != hipSuccess ||
        hipEventElapsedTime(kernel_ms, gq::g_prof_events[slot][0], gq::g_prof_events[slot][1]) != hipSuccess)
        return gq::fail(GQ_ERR_HIP, "gq_profile_read: the armed launch did not happen or has not completed");
    return GQ_OK;
}

// ---- a captured graph's kernel nodes as plain stream launches (gq_launch_plan_*) ----------------------------------------------
// A replayed HIP graph pays a boundary of its own between two replays -- 3.9-4.8 us for the two-kernel ResNet-50 step (62.4 us per
// replay against 57.6 us for the same two launches issued directly: tools/direct_vs_graph.py) -- where two launches on a stream have
// none.  The quantizer captures its steps ONCE with stream capture (the capture is what collects the launches and their arguments
// without a line of per-configuration code) and then replays the captured kernel nodes as plain launches: the node's function,
// grid, block and argument pointers are read back from the graph, which stays alive as the owner of the argument storage.
namespace gq {
struct LaunchPlan {
    std::vector<hipKernelNodeParams> nodes;
};
}  // namespace gq

GQ_API int gq_launch_plan_create(void *hip_graph, void **plan, int *nodes) {
    if (!hip_graph || !plan) return gq::fail(GQ_ERR_INVALID_ARG, "gq_launch_plan_create: null pointer");
    *plan = nullptr;
    hipGraph_t g = static_cast<hipGraph_t>(hip_graph);
    size_t n = 0, nroot = 0;
    if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess || n == 0 || n > 64)
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_launch_plan_create: the graph has %zu nodes (1 .. 64 served)", n);
    if (hipGraphGetRootNodes(g, nullptr, &nroot) != hipSuccess || nroot != 1)
        return gq::fail(GQ_ERR_UNSUPPORTED, "gq_launch_plan_create: the graph has %zu root nodes: not one chain of launches", nroot);
    hipGraphNode_t node = nullptr;
    if (hipGraphGetRootNodes(g, &node, &nroot) != hipSuccess) return gq::fail(GQ_ERR_HIP, "gq_launch_plan_create: hipGraphGetRootNodes");
    auto *p = new gq::LaunchPlan();
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        hipKernelNodeParams kp;
        size_t nd = 0;
        if (hipGraphNodeGetType(node, &type) != hipSuccess || type != hipGraphNodeTypeKernel ||
            hipGraphKernelNodeGetParams(node, &kp) != hipSuccess || !kp.func || (!kp.kernelParams && !kp.extra)) {
            delete p;
            return gq::fail(GQ_ERR_UNSUPPORTED, "gq_launch_plan_create: node %zu is not a kernel launch with readable parameters", i);
        }
        p->nodes.push_back(kp);
        if (hipGraphNodeGetDependentNodes(node, nullptr, &nd) != hipSuccess || nd != (i + 1 < n ? 1u : 0u)) {
            delete p;
            return gq::fail(GQ_ERR_UNSUPPORTED, "gq_launch_plan_create: node %zu has %zu dependents: not one chain of launches", i, nd);
        }
        if (nd == 1 && hipGraphNodeGetDependentNodes(node, &node, &nd) != hipSuccess) {
            delete p;
            return gq::fail(GQ_ERR_HIP, "gq_launch_plan_create: hipGraphNodeGetDependentNodes");
        }
    }
    if (nodes) *nodes = (int)n;
    *plan = p;
    return GQ_OK;
}

GQ_API int gq_launch_plan_run(void *plan, void *stream) {
    if (!plan) return gq::fail(GQ_ERR_INVALID_ARG, "gq_launch_plan_run: null plan");
    hipStream_t st = gq::as_stream(stream);
    for (const hipKernelNodeParams &kp : static_cast<gq::LaunchPlan *>(plan)->nodes) {
        hipError_t e;
        if (kp.kernelParams) {
            e = hipLaunchKernel(kp.func, kp.gridDim, kp.blockDim, kp.kernelParams, kp.sharedMemBytes, st);
        } else {      // (a launch captured with an argument buffer instead of an array of pointers)
            e = hipModuleLaunchKernel(static_cast<hipFunction_t>(kp.func), kp.gridDim.x, kp.gridDim.y, kp.gridDim.z, kp.blockDim.x,
                                      kp.blockDim.y, kp.blockDim.z, kp.sharedMemBytes, st, nullptr, kp.extra);
        }
        if (e != hipSuccess) return gq::fail(GQ_ERR_HIP, "gq_launch_plan_run: %s", hipGetErrorString(e));
    }
    return GQ_OK;
}

GQ_API void gq_launch_plan_destroy(void *plan) { delete static_cast<gq::LaunchPlan *>(plan); }

