// crp_rccl.h -- RCCL, loaded with dlopen on first use (crp_comm.cpp): a single-GPU process never pays for it, and the
// library has no link-time dependency on librccl.so.  Shared by the two exchanges of the path: crp_comm.cpp (one process
// per GPU, ncclCommInitRank) and crp_node.cpp (one process over N GPUs, ncclCommInitAll).
#pragma once
#include <string>

#include <rccl/rccl.h>

namespace crp {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

const Rccl *rccl();                    // nullptr when librccl.so cannot be loaded (rccl_load_error says why)
const std::string &rccl_load_error();

}  // namespace crp
