// Minimal device-runtime helpers for FFI hosts that have no HIP binding of
// their own (ctypes tests, bench.py, a cgo/JNI caller): memory, streams,
// events.  Thin, checked wrappers -- no policy.
#include <hip/hip_runtime.h>

#include "acm_internal.h"

extern "C" int acm_rt_set_device(int device)
{
	ACM_HIP_TRY(hipSetDevice(device));
	return ACM_OK;
}

extern "C" int acm_rt_malloc(void **out, size_t bytes)
{
	if (!out)
		return acm::fail(ACM_ERR_ARG, "acm_rt_malloc: null out");
	*out = nullptr;
	hipError_t e = hipMalloc(out, bytes ? bytes : 16);
	if (e == hipErrorOutOfMemory)
		return acm::fail(ACM_ERR_NOMEM, "acm_rt_malloc: %zu bytes: out of device memory", bytes);
	ACM_HIP_TRY(e);
	return ACM_OK;
}

extern "C" int acm_rt_free(void *p)
{
	if (p)
		ACM_HIP_TRY(hipFree(p));
	return ACM_OK;
}

extern "C" int acm_rt_host_alloc(void **out, size_t bytes)
{
	if (!out)
		return acm::fail(ACM_ERR_ARG, "acm_rt_host_alloc: null out");
	ACM_HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
	return ACM_OK;
}

extern "C" int acm_rt_host_free(void *p)
{
	if (p)
		ACM_HIP_TRY(hipHostFree(p));
	return ACM_OK;
}

extern "C" int acm_rt_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
	ACM_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
	ACM_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
	ACM_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_memset(void *dst, int value, size_t bytes, void *stream)
{
	ACM_HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_stream_create(void **out)
{
	hipStream_t s;
	ACM_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	*out = (void *)s;
	return ACM_OK;
}

extern "C" int acm_rt_stream_destroy(void *stream)
{
	if (stream)
		ACM_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_stream_sync(void *stream)
{
	ACM_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_device_sync(void)
{
	ACM_HIP_TRY(hipDeviceSynchronize());
	return ACM_OK;
}

extern "C" int acm_rt_event_create(void **out)
{
	hipEvent_t e;
	ACM_HIP_TRY(hipEventCreate(&e));
	*out = (void *)e;
	return ACM_OK;
}

extern "C" int acm_rt_event_destroy(void *ev)
{
	if (ev)
		ACM_HIP_TRY(hipEventDestroy((hipEvent_t)ev));
	return ACM_OK;
}

extern "C" int acm_rt_event_record(void *ev, void *stream)
{
	ACM_HIP_TRY(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
	return ACM_OK;
}

extern "C" int acm_rt_event_sync(void *ev)
{
	ACM_HIP_TRY(hipEventSynchronize((hipEvent_t)ev));
	return ACM_OK;
}

extern "C" int acm_rt_event_elapsed_ms(void *start, void *stop, float *ms)
{
	ACM_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
	return ACM_OK;
}

extern "C" int acm_rt_device_info(int device, char *name, int name_cap, int *cus, size_t *mem_bytes,
    int *lds_per_cu)
{
	hipDeviceProp_t p;
	ACM_HIP_TRY(hipGetDeviceProperties(&p, device));
	if (name && name_cap > 0) {
		snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
	}
	if (cus) *cus = p.multiProcessorCount;
	if (mem_bytes) *mem_bytes = p.totalGlobalMem;
	if (lds_per_cu) *lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
	return ACM_OK;
}
