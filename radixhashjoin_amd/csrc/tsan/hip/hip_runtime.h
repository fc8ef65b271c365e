// A HOST-ONLY stand-in for <hip/hip_runtime.h>, for ONE purpose: compiling rhj_api.hip with g++ -fsanitize=thread
// (make -C radixhashjoin_amd/csrc tsan), so that the host threads of the drop-in call -- the six stager workers, the
// downloader, the page pre-faulters of rhj_join, and several query threads with a context each -- run under ThreadSanitizer
// on the source that ships.  GPU sanitizers are not available on this pool; the host side is plain std::thread code.
//
// Nothing here is part of the product, and nothing here computes a join: "device memory" is host memory, a stream is a
// thread that executes its queue in order (so a DMA out of a pinned buffer really does run while a worker wants to refill
// that buffer: the protocol under test), an event is a flag with release / acquire semantics.  The kernel launchers of
// rhj_internal.h are replaced by fake_hip.cpp's: they enqueue nothing but the bookkeeping the host logic reads back (a
// result count, a few pair bytes).
#pragma once
#include <cstddef>
#include <cstdint>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

struct FakeStream;
struct FakeEvent;
typedef FakeStream *hipStream_t;
typedef FakeEvent *hipEvent_t;

struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipIpcMemLazyEnablePeerAccess = 1 };
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p);
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned flags);
hipError_t hipIpcCloseMemHandle(void *p);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipSetDevice(int dev);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipMemGetInfo(size_t *free_b, size_t *total_b);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t st);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t *st, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipStreamWaitEvent(hipStream_t st, hipEvent_t ev, unsigned flags);
hipError_t hipEventCreate(hipEvent_t *ev);
hipError_t hipEventCreateWithFlags(hipEvent_t *ev, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t ev);
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t ev);
hipError_t hipEventQuery(hipEvent_t ev);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
