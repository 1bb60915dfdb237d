// Peer-to-peer mailbox transport of the per-pass exchanges of the sharded PCG (see p2p.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include "comm.hpp"
#include "../device/p2p_device.hpp"

namespace i3d {

P2PLayout p2p_layout(int world, int red_cap /* doubles per all-reduce */, int halo_cap /* rim entries per pair */);

struct P2PEngine {
    int rank = 0, world = 1; bool ready = false;
    P2PLayout L{};
    unsigned char* mailbox = nullptr;                 // this rank's mailbox (fine-grained device memory)
    unsigned char* peer[P2P_MAX_RANKS] = {nullptr};   // every rank's mailbox as seen from this device (peer[rank] == mailbox)
    bool opened[P2P_MAX_RANKS] = {false};
    int* d_err = nullptr; int* d_lists = nullptr;
    unsigned long long* d_epoch_red = nullptr;        // all-reduces performed so far (device counter, see p2p_device.hpp)
    unsigned long long epoch_halo = 0;
    unsigned long long spin_limit = P2P_SPIN_LIMIT;   // bound of every wait (shader-clock ticks)
    int wg_cap = 0;                                   // > 0: cap on the workgroups of the multi-workgroup exchange kernels (rank simulation on one device)

    int  create(int rank, int world, int red_cap, int halo_cap);
    int  export_handle(void* out64);                  // hipIpcMemHandle_t of the mailbox
    int  attach_ipc(const void* handles);             // world x 64 bytes, in rank order (one process per GPU)
    void attach_pointer(int k, unsigned char* p);     // same process (rank simulation)
    void destroy();
    P2PDev device() const;                            // handle for kernels that reduce in place (p2p_allreduce_wg)
    int  allreduce(double* dev, size_t n, hipStream_t st);
    int  set_halo_lists(const HaloPlan& h, hipStream_t st);      // 0 ok, 2 = a pair's rim exceeds the mailbox, 1 = HIP error
    int  push_halo(float* vec, const HaloPlan& h, hipStream_t st);
    int  check(hipStream_t st);                       // 1 if a spin timed out
    int  selftest_fused(hipStream_t st);              // the in-kernel, multi-workgroup exchanges of pcg_fused.hip with known values (all ranks call it together); 0 = ok
};

}  // namespace i3d
