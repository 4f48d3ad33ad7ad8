/* HIP runtime calls behind a C face; see hip_shim.h.  No kernels here. */
#include <hip/hip_runtime.h>

#include "hip_shim.h"

extern "C" {

int hufs_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int hufs_get_device(int *device) {
    return (int)hipGetDevice(device);
}

int hufs_set_device(int device) {
    return (int)hipSetDevice(device);
}

const char *hufs_error_string(int error) {
    return hipGetErrorString((hipError_t)error);
}

void *hufs_malloc(size_t size) {
    void *p = NULL;
    if (hipMalloc(&p, size ? size : 1) != hipSuccess) {
        return NULL;
    }
    return p;
}

void hufs_free(void *ptr) {
    if (ptr) {
        (void)hipFree(ptr);
    }
}

void *hufs_host_alloc(size_t size) {
    void *p = NULL;
    if (hipHostMalloc(&p, size ? size : 1, 0) != hipSuccess) {
        return NULL;
    }
    return p;
}

void hufs_host_free(void *ptr) {
    if (ptr) {
        (void)hipHostFree(ptr);
    }
}

int hufs_copy_h2d(void *dst, const void *src, size_t size, void *stream) {
    if (size == 0) {
        return 0;
    }
    return (int)hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, (hipStream_t)stream);
}

int hufs_copy_d2h(void *dst, const void *src, size_t size, void *stream) {
    if (size == 0) {
        return 0;
    }
    return (int)hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, (hipStream_t)stream);
}

int hufs_memset(void *dst, int byte, size_t size, void *stream) {
    if (size == 0) {
        return 0;
    }
    return (int)hipMemsetAsync(dst, byte, size, (hipStream_t)stream);
}

int hufs_stream_create(void **stream) {
    hipStream_t s = NULL;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *stream = (void *)s;
    return (int)e;
}

int hufs_stream_destroy(void *stream) {
    return stream ? (int)hipStreamDestroy((hipStream_t)stream) : 0;
}

int hufs_device_sync(void) {
    return (int)hipDeviceSynchronize();
}

int hufs_stream_sync(void *stream) {
    return (int)hipStreamSynchronize((hipStream_t)stream);
}

void *hufs_event_create(void) {
    hipEvent_t e = NULL;
    if (hipEventCreate(&e) != hipSuccess) {
        return NULL;
    }
    return (void *)e;
}

void hufs_event_destroy(void *event) {
    if (event) {
        (void)hipEventDestroy((hipEvent_t)event);
    }
}

int hufs_event_record(void *event, void *stream) {
    return (int)hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
}

void *hufs_event_create_untimed(void) {
    hipEvent_t e = nullptr;
    return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess ? (void *)e : nullptr;
}

int hufs_event_sync(void *event) {
    return (int)hipEventSynchronize((hipEvent_t)event);
}

int hufs_event_elapsed_ms(void *start, void *stop, float *ms) {
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e != hipSuccess) {
        return (int)e;
    }
    return (int)hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
}

} /* extern "C" */
