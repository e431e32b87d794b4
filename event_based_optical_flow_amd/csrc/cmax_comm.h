// RCCL binding of libcmax_hip.so (library-internal).
//
// The collectives of the time-sliced multi-GPU objective are enqueued by the library itself,
// on the same stream as the event kernels: vote -> all-reduce(images) -> contrast + gather -> all-reduce(gradient).
// RCCL is bound at run time (dlopen of the librccl the process already holds -- torch's -- else the ROCm one), so a
// single-GPU process never loads it and the shared object has no link-time dependency on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace cmax {

struct Comm;  // one RCCL communicator (one rank = one process = one GPU)

enum CommType { kCommF32 = 0, kCommF64 = 1 };
enum CommOp { kCommSum = 0, kCommMin = 1, kCommMax = 2 };

__attribute__((visibility("hidden"))) int comm_unique_id(void *id128_host);
__attribute__((visibility("hidden"))) int comm_create(const void *id128_host, int nranks, int rank, Comm **out);
__attribute__((visibility("hidden"))) void comm_destroy(Comm *c);
__attribute__((visibility("hidden"))) int comm_nranks(const Comm *c);
__attribute__((visibility("hidden"))) int comm_rank(const Comm *c);
__attribute__((visibility("hidden"))) int comm_version();
// 0 if RCCL can be bound in this process (binds it), else CMAX_ENODEV; path_out: the shared object the entry points came from
__attribute__((visibility("hidden"))) int comm_available(char *path_out, int cap);
// in-place all-reduce of `count` elements at `buf` (device), enqueued on `s`
__attribute__((visibility("hidden"))) int comm_allreduce(Comm *c, void *buf, size_t count, CommType type, CommOp op, hipStream_t s);
// several in-place all-reduces as ONE grouped RCCL call (ncclGroupStart / End): one launch, one synchronisation of the ranks
__attribute__((visibility("hidden"))) int comm_allreduce_group(Comm *c, void *const *bufs, const size_t *counts, const CommType *types, int n,
                                                             CommOp op, hipStream_t s);

}  // namespace cmax
