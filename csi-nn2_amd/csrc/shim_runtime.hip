// shim_runtime.hip -- device, memory, stream, event and graph entry points of the C-ABI
// (include/shl_mi355x.h).  Thin, allocation-free wrappers over the HIP runtime so that the C
// host backend never includes a HIP header.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

namespace shl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    (void)hipGetLastError();  // clear the sticky error
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
        return SHL_MI355X_ENODEV;
    if (e == hipErrorOutOfMemory) return SHL_MI355X_ENOMEM;
    return SHL_MI355X_EHIP;
}

}  // namespace shl

using shl::hip_fail;
using shl::set_error;

extern "C" {

int shl_mi355x_abi_version(void) { return SHL_MI355X_ABI_VERSION; }

const char *shl_mi355x_last_error(void) { return shl::g_err; }

int shl_mi355x_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        hip_fail(e, "hipGetDeviceCount");
        return 0;
    }
    return n;
}

int shl_mi355x_set_device(int ordinal)
{
    SHL_HIP(hipSetDevice(ordinal));
    return SHL_MI355X_OK;
}

int shl_mi355x_device_info(char *arch, size_t arch_len, int32_t *cu_count, int64_t *hbm_bytes)
{
    int dev = 0;
    SHL_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SHL_HIP(hipGetDeviceProperties(&prop, dev));
    if (arch && arch_len) {
        strncpy(arch, prop.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return SHL_MI355X_OK;
}

int shl_mi355x_device_bus_id(char *buf, size_t len)
{
    if (!buf || len < 14) {
        shl::set_error("device_bus_id: need a buffer of at least 14 bytes");
        return SHL_MI355X_EINVAL;
    }
    int dev = 0;
    SHL_HIP(hipGetDevice(&dev));
    SHL_HIP(hipDeviceGetPCIBusId(buf, (int)len, dev));
    return SHL_MI355X_OK;
}

void *shl_mi355x_malloc(size_t bytes)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc");
        return nullptr;
    }
    return p;
}

int shl_mi355x_free(void *ptr_dev)
{
    if (ptr_dev) SHL_HIP(hipFree(ptr_dev));
    return SHL_MI355X_OK;
}

int shl_mi355x_is_device_ptr(const void *ptr)
{
    if (ptr == nullptr) return 0;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc'ed memory reports an error: that means "host"
        return 0;
    }
    return attr.type == hipMemoryTypeDevice ? 1 : 0;
}

int shl_mi355x_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
    if (bytes == 0) return SHL_MI355X_OK;
    SHL_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_download(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
    if (bytes == 0) return SHL_MI355X_OK;
    SHL_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream)
{
    if (bytes == 0) return SHL_MI355X_OK;
    SHL_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_memset(void *dst_dev, int byte_value, size_t bytes, void *stream)
{
    if (bytes == 0) return SHL_MI355X_OK;
    SHL_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

void *shl_mi355x_stream_create(void)
{
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        hip_fail(e, "hipStreamCreateWithFlags");
        return nullptr;
    }
    return s;
}

int shl_mi355x_stream_destroy(void *stream)
{
    if (stream) SHL_HIP(hipStreamDestroy((hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_stream_sync(void *stream)
{
    SHL_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SHL_MI355X_OK;
}

void *shl_mi355x_event_create(void)
{
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) {
        hip_fail(e, "hipEventCreate");
        return nullptr;
    }
    return ev;
}

int shl_mi355x_event_destroy(void *event)
{
    if (event) SHL_HIP(hipEventDestroy((hipEvent_t)event));
    return SHL_MI355X_OK;
}

int shl_mi355x_event_record(void *event, void *stream)
{
    SHL_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_event_elapsed_ms(void *start, void *stop, float *ms)
{
    SHL_HIP(hipEventSynchronize((hipEvent_t)stop));
    SHL_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SHL_MI355X_OK;
}

int shl_mi355x_graph_begin(void *stream)
{
    SHL_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return SHL_MI355X_OK;
}

void *shl_mi355x_graph_end(void *stream)
{
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (e != hipSuccess) {
        hip_fail(e, "hipStreamEndCapture");
        return nullptr;
    }
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
        hip_fail(e, "hipGraphInstantiate");
        return nullptr;
    }
    return exec;
}

int shl_mi355x_graph_launch(void *graph_exec, void *stream)
{
    SHL_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return SHL_MI355X_OK;
}

int shl_mi355x_graph_destroy(void *graph_exec)
{
    if (graph_exec) SHL_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return SHL_MI355X_OK;
}

}  // extern "C"
