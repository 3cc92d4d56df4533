#include "rccl_dyn.hpp"

#include <dlfcn.h>

#include <mutex>
#include <string>

#include "bmx_common.hpp"

namespace bmx {
namespace rccl {

Api& api() {
    static Api a;
    return a;
}

void load(const char* path) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    Api& a = api();
    if (a.ready()) return;
    void* h = (path && *path) ? dlopen(path, RTLD_NOW | RTLD_GLOBAL) : dlopen(nullptr, RTLD_NOW);
    if (!h) throw Error(BMX_ERR_EXCHANGE, std::string("cannot load RCCL: ") + dlerror());
    auto sym = [&](const char* name) {
        void* p = dlsym(h, name);
        if (!p) throw Error(BMX_ERR_EXCHANGE, std::string("RCCL symbol missing: ") + name);
        return p;
    };
    a.GetUniqueId = reinterpret_cast<int (*)(UniqueId*)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<int (*)(Comm*, int, UniqueId, int)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<int (*)(Comm)>(sym("ncclCommDestroy"));
    a.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, Comm, hipStream_t)>(sym("ncclAllGather"));
    a.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
    a.GroupStart = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
}

}  // namespace rccl
}  // namespace bmx
