// RCCL entry points resolved at run time from an already-loaded RCCL (the one torch ships in a multi-GPU process):
// the library has no link-time dependency on RCCL and a single-GPU user never touches it.
// Only what the engine's one exchange step needs: communicator set-up and an in-place all-gather on the engine's
// stream (RCCL over xGMI in production).  The declarations mirror rccl.h (ncclUniqueId = 128 opaque bytes passed by
// value, ncclUint8 = 1, ncclSuccess = 0).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace bmx {
namespace rccl {

struct UniqueId {
    char internal[128];
};
typedef void* Comm;

struct Api {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;  // optional: two gathers of one search go out as one launch
    int (*GroupEnd)() = nullptr;
    bool ready() const { return GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};

Api& api();                  // process-wide table
void load(const char* path); // dlopen(path) (or the global scope when path is null / empty) and fill the table; throws

}  // namespace rccl
}  // namespace bmx
