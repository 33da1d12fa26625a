// Multi-GPU exchange of the render path: the samplings of renderer.rs:32-43 are sharded by index over the
// ranks, every rank accumulates its share, and ONE all-reduce (sum) of the fp32 accumulators over RCCL precedes the resolve
// of renderer.rs:64-90.  RCCL (librccl.so, the ROCm build of the NCCL API) is loaded on first use: a single-GPU host never
// pays for it, and a box without it gets HR_ERR_UNSUPPORTED with the loader's message — there is no host-side fallback sum.
// Included by hr_api.hip only.
#pragma once
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <hip/hip_runtime.h>

namespace hrcomm {

typedef struct { char internal[128]; } UniqueId;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                                // ncclComm_t
enum { kFloat = 7, kSum = 0 };                     // ncclFloat32, ncclSum

struct Api {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommInitAll)(Comm *, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    // what the communicator says about itself (hr_comm_info: the evidence a bench line carries that the collective ran over N ranks)
    int (*CommCount)(Comm, int *) = nullptr;
    int (*CommUserRank)(Comm, int *) = nullptr;
    int (*CommCuDevice)(Comm, int *) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    std::string error;
    std::string path;     // the shared object ncclAllReduce was resolved from (dladdr)
    bool reused = false;  // true: an RCCL that was already mapped into the process (the host's own — e.g. PyTorch's — build) was taken
};

inline Api &api() {
    static Api a;
    return a;
}
// false (and api().error set) when RCCL cannot be loaded
inline bool load() {
    Api &a = api();
    if (a.lib) return true;
    // ONE RCCL per process.  A host that has its own RCCL mapped already (PyTorch ships torch/lib/librccl.so and maps it with libtorch_hip;
    // under torch.distributed.run its NCCL backend runs on that build) must not get a second build beside it — two RCCL runtimes in one
    // process, each with its own proxy threads and IPC handles, is the kind of thing that only fails on the first real 8-GPU run.  So:
    // look for a mapped object called librccl* first and take THAT (RTLD_NOLOAD: a handle to what is there, nothing new is loaded).
    {
        std::string found;
        dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *data) -> int {
            const char *nm = info->dlpi_name;
            if (nm && strstr(nm, "librccl")) { *static_cast<std::string *>(data) = nm; return 1; }
            return 0;
        }, &found);
        if (!found.empty()) {
            a.lib = dlopen(found.c_str(), RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            a.reused = a.lib != nullptr;
        }
    }
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        if (a.lib) break;
        a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!a.lib) { a.error = std::string("cannot load RCCL: ") + dlerror(); return false; }
    bool ok = true;
    auto sym = [&](const char *name) -> void * {
        void *p = dlsym(a.lib, name);
        if (!p) { ok = false; a.error = std::string("RCCL symbol missing: ") + name; }
        return p;
    };
    a.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
    a.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
    a.CommInitAll = (int (*)(Comm *, int, const int *))sym("ncclCommInitAll");
    a.AllReduce = (int (*)(const void *, void *, size_t, int, int, Comm, hipStream_t))sym("ncclAllReduce");
    a.GroupStart = (int (*)())sym("ncclGroupStart");
    a.GroupEnd = (int (*)())sym("ncclGroupEnd");
    a.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
    a.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    a.CommCount = (int (*)(Comm, int *))sym("ncclCommCount");
    a.CommUserRank = (int (*)(Comm, int *))sym("ncclCommUserRank");
    a.CommCuDevice = (int (*)(Comm, int *))sym("ncclCommCuDevice");
    a.GetVersion = (int (*)(int *))sym("ncclGetVersion");
    if (!ok) { dlclose(a.lib); a.lib = nullptr; return false; }
    Dl_info di;
    if (dladdr((void *)a.AllReduce, &di) && di.dli_fname) a.path = di.dli_fname;
    return ok;
}

}  // namespace hrcomm
