// rccl_stub.cpp -- TEST-ONLY stand-in for librccl (loaded through LDPC_HIP_RCCL_PATH by tests/test_gpu_chain.py).
//
// The GPU pool gives one GPU per box, and real RCCL refuses one device twice in a communicator, so the N > 1 communicator code of
// csrc/ldpc_multi.hpp (cached ncclCommInitAll, the grouped all-reduce, the error path) could never run there.  This library
// implements the six entry points the layer resolves, over host memory, with the checks a real multi-rank collective implies:
//   * ncclAllReduce outside ncclGroupStart/End with more than one rank in the communicator is refused (one thread drives all
//     ranks: ungrouped, the first call would block for ever in the real library);
//   * at ncclGroupEnd every communicator of a group must have been given exactly one call, with equal count / type / op --
//     a rank that is missing is reported as an error (the real library would hang), which is what the failure-injection test needs.
// It links the same libamdhip64 as libldpc_hip.so (the test driver is a plain C program, no torch in the process).
//
//   g++ -O1 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/cpp/rccl_stub.cpp -o librccl_stub.so -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Group { int n = 0; };
struct Op { const void *send; void *recv; size_t count; ncclDataType_t type; ncclRedOp_t op; hipStream_t stream; };

}  // namespace

struct ncclComm {
    std::shared_ptr<Group> group;
    int rank = 0, device = 0;
};

namespace {

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_ops;
std::mutex g_mu;
long long g_allreduce_groups = 0, g_inits = 0;

ncclResult_t execute(std::vector<std::pair<ncclComm *, Op>> &ops) {
    // sort the calls by communicator group; each group must be complete
    std::map<Group *, std::vector<std::pair<ncclComm *, Op>>> by;
    for (auto &o : ops) by[o.first->group.get()].push_back(o);
    for (auto &kv : by) {
        auto &v = kv.second;
        if ((int)v.size() != kv.first->n) return ncclInvalidUsage;   // a rank is missing: the real collective would never complete
        std::vector<char> seen((size_t)kv.first->n, 0);
        for (auto &o : v) {
            if (seen[(size_t)o.first->rank]) return ncclInvalidUsage;
            seen[(size_t)o.first->rank] = 1;
            if (o.second.count != v[0].second.count || o.second.type != v[0].second.type || o.second.op != v[0].second.op) return ncclInvalidArgument;
        }
        if (v[0].second.type != ncclUint64 || v[0].second.op != ncclSum) return ncclInvalidArgument;   // all this layer uses
        const size_t cnt = v[0].second.count;
        std::vector<unsigned long long> sum(cnt, 0ull), tmp(cnt);
        for (auto &o : v) {   // a collective enqueued on a stream runs after what the stream holds
            if (hipSetDevice(o.first->device) != hipSuccess) return ncclUnhandledCudaError;
            if (hipStreamSynchronize(o.second.stream) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpy(tmp.data(), o.second.send, cnt * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            for (size_t i = 0; i < cnt; ++i) sum[i] += tmp[i];
        }
        for (auto &o : v) {
            if (hipSetDevice(o.first->device) != hipSuccess) return ncclUnhandledCudaError;
            if (hipMemcpy(o.second.recv, sum.data(), cnt * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        }
        std::lock_guard<std::mutex> lock(g_mu);
        ++g_allreduce_groups;
    }
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
    if (!comm || ndev < 1) return ncclInvalidArgument;
    auto g = std::make_shared<Group>();
    g->n = ndev;
    for (int i = 0; i < ndev; ++i) {
        comm[i] = new ncclComm();
        comm[i]->group = g; comm[i]->rank = i; comm[i]->device = devlist ? devlist[i] : i;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    ++g_inits;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<std::pair<ncclComm *, Op>> ops;
    ops.swap(g_ops);
    return execute(ops);
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff) return ncclInvalidArgument;
    const Op o{sendbuff, recvbuff, count, datatype, op, stream};
    if (g_depth > 0) { g_ops.emplace_back(comm, o); return ncclSuccess; }
    if (comm->group->n != 1) return ncclInvalidUsage;   // one thread, several ranks, no group: would block for ever
    std::vector<std::pair<ncclComm *, Op>> one{{comm, o}};
    return execute(one);
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidUsage: return "invalid usage (stub: incomplete or ungrouped collective)";
    case ncclInvalidArgument: return "invalid argument";
    case ncclUnhandledCudaError: return "unhandled HIP error";
    default: return "error";
    }
}

// test hooks
long long ldpc_rccl_stub_allreduces(void) { std::lock_guard<std::mutex> lock(g_mu); return g_allreduce_groups; }
long long ldpc_rccl_stub_inits(void) { std::lock_guard<std::mutex> lock(g_mu); return g_inits; }

}  // extern "C"
