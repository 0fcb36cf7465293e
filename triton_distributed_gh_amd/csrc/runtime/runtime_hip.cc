// Host runtime for the symmetric heap, stream-ordered signals and copy-engine transfers.
//
// Replaces the reference's `pyrocshmem` extension
// (reference: shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:101-434) and the stream-op helpers
// (python/triton_dist/kernels/amd/common_ops.py:72-110).  Differences by design:
//   * one arena per rank is allocated and IPC-mapped ONCE; symmetric tensors are bump-allocated
//     at identical offsets, so no per-tensor handle exchange is needed;
//   * a second, small fine-grained arena holds flags so system-scope atomics are coherent on xGMI;
//   * handle exchange itself is done by the Python side over torch.distributed (gloo/RCCL),
//     this file only produces / opens the 64-byte handles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <string>

#include "tdist/common.hpp"

static thread_local std::string g_last_error;

extern "C" void tdist_set_last_error(const char* what, int code, const char* file, int line) {
  char buf[1024];
  const char* hip_msg = code > 0 ? hipGetErrorString((hipError_t)code) : "";
  snprintf(buf, sizeof(buf), "%s failed (code %d %s) at %s:%d", what, code, hip_msg, file, line);
  g_last_error = buf;
}

TDIST_API const char* tdist_last_error() { return g_last_error.c_str(); }

TDIST_API int tdist_device_count(int* n) {
  TDIST_HIP_TRY(hipGetDeviceCount(n));
  return 0;
}

TDIST_API int tdist_set_device(int dev) {
  TDIST_HIP_TRY(hipSetDevice(dev));
  return 0;
}

// out[0]=multiProcessorCount out[1]=maxSharedMemoryPerMultiProcessor out[2]=clockRate(kHz)
// out[3]=totalGlobalMem (MiB) out[4]=warpSize out[5]=l2CacheSize
TDIST_API int tdist_device_props(int dev, long long* out, char* arch_name, int arch_len) {
  hipDeviceProp_t p;
  TDIST_HIP_TRY(hipGetDeviceProperties(&p, dev));
  out[0] = p.multiProcessorCount;
  out[1] = (long long)p.maxSharedMemoryPerMultiProcessor;
  out[2] = p.clockRate;
  out[3] = (long long)(p.totalGlobalMem >> 20);
  out[4] = p.warpSize;
  out[5] = p.l2CacheSize;
  if (arch_name && arch_len > 0) {
    strncpy(arch_name, p.gcnArchName, arch_len - 1);
    arch_name[arch_len - 1] = 0;
  }
  return 0;
}

// mode: 0 = coarse-grained hipMalloc, 1 = fine-grained, 2 = uncached.
TDIST_API int tdist_malloc(void** out, size_t bytes, int mode) {
  if (mode == 1) {
    TDIST_HIP_TRY(hipExtMallocWithFlags(out, bytes, hipDeviceMallocFinegrained));
  } else if (mode == 2) {
    TDIST_HIP_TRY(hipExtMallocWithFlags(out, bytes, hipDeviceMallocUncached));
  } else {
    TDIST_HIP_TRY(hipMalloc(out, bytes));
  }
  return 0;
}

TDIST_API int tdist_free(void* p) {
  TDIST_HIP_TRY(hipFree(p));
  return 0;
}

TDIST_API int tdist_memset(void* p, int value, size_t bytes) {
  TDIST_HIP_TRY(hipMemset(p, value, bytes));
  TDIST_HIP_TRY(hipDeviceSynchronize());
  return 0;
}

TDIST_API int tdist_memset_async(void* p, int value, size_t bytes, void* stream) {
  TDIST_HIP_TRY(hipMemsetAsync(p, value, bytes, (hipStream_t)stream));
  return 0;
}

TDIST_API int tdist_ipc_handle_size() { return (int)sizeof(hipIpcMemHandle_t); }

TDIST_API int tdist_ipc_get_handle(void* p, void* handle_out) {
  hipIpcMemHandle_t h;
  TDIST_HIP_TRY(hipIpcGetMemHandle(&h, p));
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

TDIST_API int tdist_ipc_open_handle(const void* handle_in, void** out) {
  hipIpcMemHandle_t h;
  memcpy(&h, handle_in, sizeof(h));
  TDIST_HIP_TRY(hipIpcOpenMemHandle(out, h, hipIpcMemLazyEnablePeerAccess));
  return 0;
}

TDIST_API int tdist_ipc_close_handle(void* p) {
  TDIST_HIP_TRY(hipIpcCloseMemHandle(p));
  return 0;
}

TDIST_API int tdist_can_access_peer(int dev, int peer, int* out) {
  TDIST_HIP_TRY(hipDeviceCanAccessPeer(out, dev, peer));
  return 0;
}

TDIST_API int tdist_enable_peer_access(int peer) {
  hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  if (e == hipErrorPeerAccessAlreadyEnabled) {
    (void)hipGetLastError();
    return 0;
  }
  TDIST_HIP_TRY(e);
  return 0;
}

// attr: 0 = performance rank, 1 = access supported, 2 = native atomics supported
TDIST_API int tdist_p2p_attribute(int attr, int src, int dst, int* out) {
  hipDeviceP2PAttr a = attr == 0 ? hipDevP2PAttrPerformanceRank
                                 : (attr == 1 ? hipDevP2PAttrAccessSupported : hipDevP2PAttrNativeAtomicSupported);
  TDIST_HIP_TRY(hipDeviceGetP2PAttribute(out, a, src, dst));
  return 0;
}

// Copy `bytes` device-to-device.  use_copy_engine != 0 asks for the SDMA path that never occupies CUs.
TDIST_API int tdist_memcpy_async(void* dst, const void* src, size_t bytes, int use_copy_engine, void* stream) {
  TDIST_HIP_TRY(hipMemcpyAsync(dst, src, bytes,
                               use_copy_engine ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice,
                               (hipStream_t)stream));
  return 0;
}

TDIST_API int tdist_stream_write_value32(void* stream, void* ptr, unsigned value) {
  TDIST_HIP_TRY(hipStreamWriteValue32((hipStream_t)stream, ptr, value, 0));
  return 0;
}
TDIST_API int tdist_stream_write_value64(void* stream, void* ptr, unsigned long long value) {
  TDIST_HIP_TRY(hipStreamWriteValue64((hipStream_t)stream, ptr, value, 0));
  return 0;
}
// cmp: 0 = EQ, 1 = GEQ
TDIST_API int tdist_stream_wait_value32(void* stream, void* ptr, unsigned value, int cmp) {
  TDIST_HIP_TRY(hipStreamWaitValue32((hipStream_t)stream, ptr, value,
                                     cmp == 0 ? hipStreamWaitValueEq : hipStreamWaitValueGte, 0xFFFFFFFFu));
  return 0;
}
TDIST_API int tdist_stream_wait_value64(void* stream, void* ptr, unsigned long long value, int cmp) {
  TDIST_HIP_TRY(hipStreamWaitValue64((hipStream_t)stream, ptr, value,
                                     cmp == 0 ? hipStreamWaitValueEq : hipStreamWaitValueGte,
                                     0xFFFFFFFFFFFFFFFFull));
  return 0;
}

TDIST_API int tdist_stream_create(void** out, int priority) {
  hipStream_t s;
  TDIST_HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
  *out = (void*)s;
  return 0;
}
TDIST_API int tdist_stream_destroy(void* s) {
  TDIST_HIP_TRY(hipStreamDestroy((hipStream_t)s));
  return 0;
}
TDIST_API int tdist_stream_sync(void* s) {
  TDIST_HIP_TRY(hipStreamSynchronize((hipStream_t)s));
  return 0;
}
TDIST_API int tdist_device_sync() {
  TDIST_HIP_TRY(hipDeviceSynchronize());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Full-mesh push all-gather on the copy engines.
//
// For every peer p != rank (visited in the order rank+1, rank+2, ... so the 7 xGMI links of a
// node are driven concurrently) and every chunk c of the local shard:
//     peer_ws[p] + (rank*rows_per_rank + c*rows_per_chunk) * row_bytes  <-  local + c*rows_per_chunk*row_bytes
//     peer_flags[p][rank*chunks + c]                                    <-  flag_value   (4-byte copy from `flag_src`)
// Data copy and flag copy are issued on the same stream, so stream order is the release edge
// (protocol of reference python/triton_dist/kernels/amd/all_gather_gemm.py:40-99).
// One C call replaces (w-1)*chunks*2 Python-level driver calls.
// ---------------------------------------------------------------------------------------------
TDIST_API int tdist_ag_push_full_mesh(int rank, int num_ranks, const void* local, const uint64_t* peer_ws,
                                      const uint64_t* peer_flags, const void* flag_src, long long rows_per_rank,
                                      long long rows_per_chunk, long long row_bytes, void* const* streams,
                                      int num_streams, int use_copy_engine, int flag_mode,
                                      unsigned flag_value) {
  TDIST_REQUIRE(rows_per_chunk > 0 && rows_per_rank % rows_per_chunk == 0, "rows_per_rank % rows_per_chunk != 0");
  TDIST_REQUIRE(num_streams > 0, "need at least one stream");
  const long long chunks = rows_per_rank / rows_per_chunk;
  const hipMemcpyKind kind = use_copy_engine ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice;
  for (int i = 1; i < num_ranks; ++i) {
    const int peer = (rank + i) % num_ranks;
    hipStream_t s = (hipStream_t)streams[(i - 1) % num_streams];
    for (long long c = 0; c < chunks; ++c) {
      const char* src = (const char*)local + c * rows_per_chunk * row_bytes;
      char* dst = (char*)peer_ws[peer] + (rank * rows_per_rank + c * rows_per_chunk) * row_bytes;
      TDIST_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(rows_per_chunk * row_bytes), kind, s));
      int* flag = (int*)peer_flags[peer] + rank * chunks + c;
      if (flag_mode == 0) {
        TDIST_HIP_TRY(hipMemcpyAsync(flag, flag_src, sizeof(int), kind, s));
      } else {
        TDIST_HIP_TRY(hipStreamWriteValue32(s, flag, flag_value, 0));
      }
    }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Virtual-memory-management arena: a symmetric heap of any size.
//
// One hipIpc mapping stays below 2 GiB on this stack, so a heap sized for a 288 GB device cannot be
// ONE hipMalloc shared through hipIpc.  Here each rank reserves a contiguous virtual range per rank
// of the node (its own and one per peer), commits physical memory in chunks on demand
// (hipMemCreate), maps its own chunks at the next offset of its own range, exports each chunk as a
// dmabuf file descriptor, and maps the descriptors received from a peer at the SAME offset of that
// peer's range.  Addresses inside a range are contiguous, so a symmetric tensor may span any number
// of chunks and peer translation stays `base[peer] + offset`.  The descriptors travel between the
// processes over a Unix-domain socket (utils/vmm.py); this file only wraps the driver calls.
// Role in the reference: rocshmem_malloc on the rocSHMEM symmetric heap
// (shmem/rocshmem_bind/pyrocshmem/src/pyrocshmem.cc:101-113).
// ---------------------------------------------------------------------------------------------
static hipMemAllocationProp vmm_prop(int dev, int uncached) {
  hipMemAllocationProp prop;
  memset(&prop, 0, sizeof(prop));
  prop.type = uncached ? hipMemAllocationTypeUncached : hipMemAllocationTypePinned;
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  return prop;
}

TDIST_API int tdist_vmm_granularity(int dev, int uncached, unsigned long long* out) {
  hipMemAllocationProp prop = vmm_prop(dev, uncached);
  size_t g = 0;
  TDIST_HIP_TRY(hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended));
  *out = g;
  return 0;
}

TDIST_API int tdist_vmm_reserve(void** out, unsigned long long bytes, unsigned long long align) {
  TDIST_HIP_TRY(hipMemAddressReserve(out, bytes, align, nullptr, 0));
  return 0;
}

TDIST_API int tdist_vmm_unreserve(void* va, unsigned long long bytes) {
  TDIST_HIP_TRY(hipMemAddressFree(va, bytes));
  return 0;
}

// Physical chunk on device `dev`; *handle_out is the driver's generic allocation handle.
TDIST_API int tdist_vmm_create(int dev, int uncached, unsigned long long bytes, unsigned long long* handle_out) {
  hipMemAllocationProp prop = vmm_prop(dev, uncached);
  hipMemGenericAllocationHandle_t h;
  TDIST_HIP_TRY(hipMemCreate(&h, bytes, &prop, 0));
  *handle_out = (unsigned long long)(uintptr_t)h;
  return 0;
}

TDIST_API int tdist_vmm_export_fd(unsigned long long handle, int* fd_out) {
  int fd = -1;
  TDIST_HIP_TRY(hipMemExportToShareableHandle(&fd, (hipMemGenericAllocationHandle_t)(uintptr_t)handle,
                                              hipMemHandleTypePosixFileDescriptor, 0));
  *fd_out = fd;
  return 0;
}

TDIST_API int tdist_vmm_import_fd(int fd, unsigned long long* handle_out) {
  hipMemGenericAllocationHandle_t h;
  // (this runtime reads the descriptor THROUGH the pointer; passing the descriptor's value cast to a
  // pointer, as the CUDA driver API takes it, crashes inside the call)
  int os_handle = fd;
  TDIST_HIP_TRY(hipMemImportFromShareableHandle(&h, (void*)&os_handle, hipMemHandleTypePosixFileDescriptor));
  *handle_out = (unsigned long long)(uintptr_t)h;
  return 0;
}

// Map a chunk at `va` and give device `dev` (the caller's device) read / write access to it.
TDIST_API int tdist_vmm_map(void* va, unsigned long long bytes, unsigned long long handle, int dev) {
  TDIST_HIP_TRY(hipMemMap(va, bytes, 0, (hipMemGenericAllocationHandle_t)(uintptr_t)handle, 0));
  hipMemAccessDesc desc;
  memset(&desc, 0, sizeof(desc));
  desc.location.type = hipMemLocationTypeDevice;
  desc.location.id = dev;
  desc.flags = hipMemAccessFlagsProtReadWrite;
  TDIST_HIP_TRY(hipMemSetAccess(va, bytes, &desc, 1));
  return 0;
}

TDIST_API int tdist_vmm_unmap(void* va, unsigned long long bytes) {
  TDIST_HIP_TRY(hipMemUnmap(va, bytes));
  return 0;
}

TDIST_API int tdist_vmm_release(unsigned long long handle) {
  TDIST_HIP_TRY(hipMemRelease((hipMemGenericAllocationHandle_t)(uintptr_t)handle));
  return 0;
}
