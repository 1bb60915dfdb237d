// TSDF fusion on the device: the volume lives in an open-addressing hash table in HBM; one launch per frame
// for allocation and one for integration.  Launch wrappers of fusion_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace i3d {

constexpr unsigned long long FUSION_EMPTY = ~0ull;
constexpr int FUSION_COORD_OFFSET = 1 << 20;                 // voxel coordinates are packed as 21-bit offsets

struct FusionTable {                                         // slot-indexed SoA; the slot of a voxel never changes until the table grows
    unsigned long long* keys;                                // packed (x, y, z) or FUSION_EMPTY
    float* sdf; float* weight; uchar4* color;                // Voxel (sparse_voxel_grid.h:56-62), colour as R,G,B
    unsigned long long* rank;                                // (frame, pixel, ray step, block index) of the FIRST insertion in the reference's sequential order
    unsigned long long* crank;                               // centre state: ~0 never a block centre, 0 block expanded, else rank of the first visit (pending)
    unsigned long long mask;                                 // capacity - 1
};
struct FusionCam { float fx, fy, cx, cy; int w, h; };
struct FusionFrame {
    float voxel_size, truncation, depth_min, depth_max, weight_sample;
    float clip[6]; int use_clip;
    int bounds[6];                                           // computeFrustumBounds
    float c2w[16], w2c[16];
    unsigned long long frame;                                // index of this integrate() call
};

void launch_fusion_clear(hipStream_t st, FusionTable t);
void launch_fusion_rehash(hipStream_t st, FusionTable src, FusionTable dst);
void launch_erode(hipStream_t st, int w, int h, const float* in, int window, float max_diff, float* out);
void launch_normals(hipStream_t st, FusionCam cam, const float* depth, float thr, float* normals);
void launch_fusion_alloc(hipStream_t st, FusionTable t, FusionFrame f, FusionCam cam, const float* depth, unsigned long long limit, unsigned long long* count, int* overflow);
void launch_fusion_integrate(hipStream_t st, FusionTable t, FusionFrame f, FusionCam dcam, FusionCam ccam, const float* depth, const float* normals, const uint8_t* bgr);
// finish
void launch_fusion_occupied(hipStream_t st, FusionTable t, int* flags);
void launch_fusion_gather_rank(hipStream_t st, FusionTable t, const int* flags, const int* offsets, unsigned long long* rank, unsigned int* slot);
void launch_fusion_keys(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_sorted, int* kxyz);
void launch_fusion_positions(hipStream_t st, long long m, const unsigned int* slot_sorted, const int* order, unsigned int* visit_slot, int* pos_of_slot);
// correctSDF in a spatially sorted compact index space (see fusion_kernels.hip)
void launch_fusion_spatial_keys(hipStream_t st, FusionTable t, long long m, const unsigned int* slots, unsigned long long* skey);
void launch_fusion_compact_init(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const int* pos_of_slot, int* compact_of_slot, float* c_sdf, int* c_pos,
                                unsigned char* c_valid, unsigned char* c_touched);
void launch_fusion_build_nbr(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const int* compact_of_slot, const unsigned char* c_valid, int* nbr);
void launch_fusion_correct(hipStream_t st, FusionTable t, long long m, float voxel_size, const unsigned int* slot_c, const int* nbr, const int* c_pos, const unsigned char* c_valid,
                           const float* c_sdf, float* c_cur, unsigned char* c_upd, int* changed);
void launch_fusion_commit(hipStream_t st, long long m, const unsigned char* c_valid, const float* c_cur, const unsigned char* c_upd, float* c_sdf, unsigned char* c_touched, int* has_update);
void launch_fusion_write_back(hipStream_t st, FusionTable t, long long m, const unsigned int* slot_c, const float* c_sdf, const unsigned char* c_touched);
void launch_fusion_valid(hipStream_t st, FusionTable t, long long m, const unsigned int* visit_slot, int* flags);
void launch_fusion_export(hipStream_t st, FusionTable t, long long m, const unsigned int* visit_slot, const int* flags, const int* offsets, int* kxyz, float* sdf, float* weight, uint8_t* rgb);

}  // namespace i3d
