// wire_kernels.hip — FeatureData::fromMsg / toMsg (graph_slam_common/src/sensor_data.cpp:78-167) on the device:
// graph_slam_msgs/Feature records <-> the estimator's frame arena (descriptor rows u8, positions 3 x n f64, valid u8).
//
// One Feature record on the wire (Feature.msg, ROS 1 serialisation), stride S = 41 + 4 D bytes:
//   +0 int32 u | +4 int32 v | +8 bool is_3d | +9 float32 keypoint_strength | +13 uint32 D | +17 float32 descriptor[D]
//   | +17+4D float64 keypoint_position x, y, z
// Descriptors travel as one float per descriptor BYTE (sensor_data.cpp:93-110), so a frame is 41 + 4 D bytes per
// keypoint on the wire for 25 + D bytes of content: both directions are pure byte shuffles bound by HBM bandwidth.
// S is odd, nothing in a record is aligned: every access below is an ALIGNED dword access plus v_alignbit, so the
// loads of neighbouring lanes fall into the same cache lines and coalesce.
//   unpack: one lane per descriptor WORD of the arena (4 wire floats = 16 unaligned bytes in, one aligned dword out);
//           the keypoint's position, is_3d, u, v and the check of the record's D are spread over the same lanes.
//   pack:   one lane per aligned dword of the record stream; each of its four bytes is derived from (record, offset).
// Algorithmic bytes per keypoint: (41 + 4 D) + (D + 25) (+8 with u,v) in either direction (D = 32: 226 B).
#include "uzl_common.hpp"
#include "wire_types.hpp"

namespace uzl {

constexpr int kWireBlk = 256;

// the four bytes at byte offset `off` of a 4-byte-aligned stream
__device__ __forceinline__ uint32_t load_u32_at(const uint32_t* __restrict__ base, uint64_t off)
{
    const uint64_t w = off >> 2;
    const uint32_t sh = (uint32_t)(off & 3) * 8;
    const uint32_t lo = base[w];
    if (sh == 0) return lo;
    const uint32_t hi = base[w + 1];
    return __funnelshift_r(lo, hi, sh);
}

// `(unsigned char) val` of sensor_data.cpp:137: truncation towards zero, low eight bits of the integer (what x86
// cvttss2si + a byte move produce); values outside the int32 range and NaN give 0x80000000 there, i.e. byte 0
__device__ __forceinline__ uint32_t float_to_byte(uint32_t bits)
{
    const float f = __uint_as_float(bits);
    if (!(fabsf(f) < 2147483648.f)) return 0u;
    return (uint32_t)((int32_t)f) & 0xffu;
}

// segs has n_segs + 1 entries (the last one is a sentinel with item_begin = n_items).  The search runs once per wave on
// its first item (uniform: scalar loads); lanes past a frame boundary step forward from there.
__device__ __forceinline__ int find_segment(const WireSeg* __restrict__ segs, int n_segs, int64_t first_item, int64_t item)
{
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) {                          // last segment whose item_begin <= first_item
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].item_begin <= first_item) lo = mid; else hi = mid - 1;
    }
    while (segs[lo + 1].item_begin <= item) ++lo;
    return lo;
}

__global__ __launch_bounds__(kWireBlk) void wire_unpack_kernel(const uint32_t* __restrict__ stage, uint8_t* __restrict__ arena,
                                                               const WireSeg* __restrict__ segs, int n_segs, int64_t n_items,
                                                               int32_t* __restrict__ uv, int32_t* __restrict__ bad)
{
    const int64_t item = (int64_t)blockIdx.x * kWireBlk + threadIdx.x;
    if (item >= n_items) return;
    const int64_t wave_first = (int64_t)blockIdx.x * kWireBlk + (threadIdx.x & ~63u);
    const uint32_t f_lo = __builtin_amdgcn_readfirstlane((uint32_t)wave_first), f_hi = __builtin_amdgcn_readfirstlane((uint32_t)(wave_first >> 32));
    const WireSeg sg = segs[find_segment(segs, n_segs, (int64_t)(((uint64_t)f_hi << 32) | f_lo), item)];
    const uint32_t local = (uint32_t)(item - sg.item_begin);          // < 16384 * 127
    const uint32_t W = (uint32_t)sg.words;
    const uint32_t i = local / W;
    const uint32_t w = local - i * W;
    const uint64_t rec = sg.src_off + (uint64_t)i * sg.stride;
    const uint64_t d0 = rec + 17 + 16ull * w;
    // 16 unaligned bytes = five aligned dwords
    const uint64_t a = d0 >> 2;
    const uint32_t sh = (uint32_t)(d0 & 3) * 8;
    const uint32_t x0 = stage[a], x1 = stage[a + 1], x2 = stage[a + 2], x3 = stage[a + 3], x4 = sh ? stage[a + 4] : 0u;
    const uint32_t word = float_to_byte(__funnelshift_r(x0, x1, sh)) | float_to_byte(__funnelshift_r(x1, x2, sh)) << 8 |
                          float_to_byte(__funnelshift_r(x2, x3, sh)) << 16 | float_to_byte(__funnelshift_r(x3, x4, sh)) << 24;
    reinterpret_cast<uint32_t*>(arena + sg.desc_off)[(size_t)i * W + w] = word;
    // the keypoint's other fields are spread over its lanes: position dwords on lanes 0..5 (strided when W < 6),
    // is_3d + the descriptor-count check on the last lane, u / v on the lane before it
    const uint64_t p0 = rec + 17 + 16ull * W;
    uint32_t* pos = reinterpret_cast<uint32_t*>(arena + sg.pos_off) + (size_t)i * 6;          // column i of the 3 x n matrix
    for (uint32_t k = w; k < 6; k += W) pos[k] = load_u32_at(stage, p0 + 4 * k);
    if (w == W - 1) {
        if (load_u32_at(stage, rec + 13) != 4 * W) atomicOr(bad, 1);
        arena[sg.valid_off + i] = (load_u32_at(stage, rec + 8) & 0xffu) ? 1 : 0;          // std::vector<bool>::push_back(is_3d) (:164)
    }
    if (uv && w == (W > 1 ? W - 2 : 0)) {
        uv[2 * (sg.feat_begin + i)] = (int32_t)load_u32_at(stage, rec);
        uv[2 * (sg.feat_begin + i) + 1] = (int32_t)load_u32_at(stage, rec + 4);
    }
}

// byte q of record i of a frame (toMsg, sensor_data.cpp:78-121)
__device__ __forceinline__ uint32_t record_byte(const uint8_t* __restrict__ desc, const uint32_t* __restrict__ pos,
                                                const uint8_t* __restrict__ valid, const int32_t* __restrict__ uv, int D,
                                                size_t i, uint32_t q)
{
    if (q < 8) {                                                   // u, v
        const uint32_t v = uv ? (uint32_t)uv[2 * i + (q >> 2)] : 0u;
        return (v >> (8 * (q & 3))) & 0xffu;
    }
    if (q == 8) return valid[i] ? 1u : 0u;                          // is_3d
    if (q < 13) return (0xbf800000u >> (8 * (q - 9))) & 0xffu;      // keypoint_strength = -1 (:96)
    if (q < 17) return ((uint32_t)D >> (8 * (q - 13))) & 0xffu;     // descriptor count
    const uint32_t e = q - 17;
    if (e < 4u * (uint32_t)D) {                                     // float val = features_.at<unsigned char>(i,j) (:104)
        const uint32_t bits = __float_as_uint((float)desc[i * D + (e >> 2)]);
        return (bits >> (8 * (e & 3))) & 0xffu;
    }
    const uint32_t p = e - 4u * (uint32_t)D;                        // keypoint_position (:113-115)
    return (pos[i * 6 + (p >> 2)] >> (8 * (p & 3))) & 0xffu;
}

__global__ __launch_bounds__(kWireBlk) void wire_pack_kernel(const uint8_t* __restrict__ arena, WireSeg sg, const int32_t* __restrict__ uv,
                                                             uint32_t* __restrict__ out, int64_t n_dwords, uint64_t n_bytes)
{
    const int64_t j = (int64_t)blockIdx.x * kWireBlk + threadIdx.x;
    if (j >= n_dwords) return;
    const uint8_t* desc = arena + sg.desc_off;
    const uint32_t* pos = reinterpret_cast<const uint32_t*>(arena + sg.pos_off);
    const uint8_t* valid = arena + sg.valid_off;
    const int D = 4 * sg.words;
    const uint32_t byte = 4u * (uint32_t)j;                          // one frame: < 16384 * 549 bytes
    uint32_t i = byte / sg.stride;
    uint32_t q = byte - i * sg.stride;
    uint32_t word = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((uint64_t)byte + k < n_bytes) word |= record_byte(desc, pos, valid, uv, D, i, q) << (8 * k);
        if (++q == sg.stride) { q = 0; ++i; }
    }
    out[j] = word;
}

void launch_wire_unpack(const uint32_t* stage, uint8_t* arena, const WireSeg* segs, int n_segs, int64_t n_items, int32_t* uv, int32_t* bad,
                        hipStream_t s)
{
    if (n_items <= 0) return;
    const int64_t blocks = (n_items + kWireBlk - 1) / kWireBlk;
    hipLaunchKernelGGL(wire_unpack_kernel, dim3((unsigned)blocks), dim3(kWireBlk), 0, s, stage, arena, segs, n_segs, n_items, uv, bad);
}

void launch_wire_pack(const uint8_t* arena, const WireSeg& sg, const int32_t* uv, uint32_t* out, uint64_t n_bytes, hipStream_t s)
{
    const int64_t n_dwords = (int64_t)((n_bytes + 3) / 4);
    if (n_dwords <= 0) return;
    const int64_t blocks = (n_dwords + kWireBlk - 1) / kWireBlk;
    hipLaunchKernelGGL(wire_pack_kernel, dim3((unsigned)blocks), dim3(kWireBlk), 0, s, arena, sg, uv, out, n_dwords, n_bytes);
}

}  // namespace uzl
