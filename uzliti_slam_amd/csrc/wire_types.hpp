// wire_types.hpp — shared between wire_kernels.hip and the estimator's host side (uzl_match.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace uzl {

// one frame of a wire batch: where its Feature records start in the staging buffer and where its arrays live in the arena
struct WireSeg {
    int64_t  item_begin;     // first workgroup of this frame in the batch (wire_unpack_kernel)
    int64_t  feat_begin;     // first keypoint of this frame in the batch (row of the optional u,v output)
    uint64_t src_off;        // byte offset of record 0 in the staging buffer
    uint64_t desc_off, pos_off, valid_off;   // byte offsets into the frame arena
    uint32_t stride;         // 41 + 4 D
    int32_t  words;          // D / 4
    int32_t  n;              // keypoints
    int32_t  _pad;           // unpack: keypoints per workgroup (wire_kpb(stride))
};

int wire_kpb(uint32_t stride);
void launch_wire_unpack(const uint32_t* stage, uint8_t* arena, const WireSeg* segs, int n_segs, int64_t n_blocks, int32_t* uv, int32_t* bad,
                        hipStream_t s);
void launch_wire_pack(const uint8_t* arena, const WireSeg& sg, const int32_t* uv, uint32_t* out, uint64_t n_bytes, hipStream_t s);

}  // namespace uzl
