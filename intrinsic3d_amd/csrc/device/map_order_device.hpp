// map_order.hip: the iteration order of the reference's voxel map for n DISTINCT keys resident on the device (see there).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace i3d {
struct MapEpoch { size_t m_end; size_t nb; };      // elements [0, m_end) are in the map when the epoch ends; bucket count nb during it
// d_order[v] = insertion index of the v-th visited key; epochs from map_epochs(n) (host/map_order.hpp).  Synchronises the stream.
hipError_t map_order_device(hipStream_t st, const int* d_keys /*[n][3]*/, size_t n, const MapEpoch* epochs, int n_epochs, int* d_order);
}  // namespace i3d
