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
//   unpack: a workgroup stages its byte range of the record stream in LDS with coalesced 16-byte loads, then one lane per
//           descriptor WORD of the arena (4 wire floats = 16 unaligned bytes from LDS in, one aligned dword out); the keypoint's
//           position, is_3d, u, v and the check of the record's D are spread over the same lanes.
//   pack:   one lane per aligned dword of the record stream; each of its four bytes is derived from (record, offset).
// Algorithmic bytes per keypoint: (41 + 4 D) + (D + 25) (+8 with u,v) in either direction (D = 32: 226 B).
#include "uzl_common.hpp"
#include "wire_types.hpp"

#include <algorithm>

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

// One workgroup = up to WireSeg::kpb keypoints of one frame (segs[k].item_begin = first WORKGROUP of frame k; the last entry is a
// sentinel).  Phase 1: the workgroup's byte range of the record stream goes to LDS with coalesced 16-byte loads (the access shape
// HBM serves at full rate; the range starts at an arbitrary byte, so it is fetched from the 16-byte boundary below it).  Phase 2:
// every lane assembles one descriptor word from five ALIGNED LDS dwords + v_alignbit and stores it coalesced; the keypoint's
// position dwords, is_3d, u / v and the check of the record's descriptor count are spread over the same lanes.
// (Measured: one lane per word straight from global memory 86 us for 231 MB; this form 74 us; the same with a register-staged
// prefetch of the next chunk and 1 - 8 chunks per workgroup 116 us: many small workgroups balance better than few pipelined ones.)
constexpr int kWireLdsBytes = 16384;    // measured on 231 MB: 8 KB 62 us, 12 KB 56, 16 KB 52, 20 KB 57, 32 KB 74 (more resident workgroups hide the stage's load latency)
__device__ __forceinline__ uint32_t lds_u32_at(const uint32_t* __restrict__ l32, uint32_t off)
{
    const uint32_t w = off >> 2, sh = (off & 3) * 8;
    const uint32_t lo = l32[w];
    if (sh == 0) return lo;
    return __funnelshift_r(lo, l32[w + 1], sh);
}

__global__ __launch_bounds__(kWireBlk) void wire_unpack_kernel(const uint4* __restrict__ stage, uint8_t* __restrict__ arena,
                                                               const WireSeg* __restrict__ segs, int n_segs,
                                                               int32_t* __restrict__ uv, int32_t* __restrict__ bad)
{
    __shared__ uint4 sraw[kWireLdsBytes / 16 + 4];
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x;
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) {                          // last frame whose first workgroup <= blk (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].item_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const WireSeg sg = segs[lo];
    const uint32_t kpb = (uint32_t)sg._pad, W = (uint32_t)sg.words, stride = sg.stride;
    const uint32_t kp0 = (uint32_t)(blk - sg.item_begin) * kpb;
    const uint32_t nkp = min(kpb, (uint32_t)sg.n - kp0);
    const uint64_t start = sg.src_off + (uint64_t)kp0 * stride, a0 = start & ~15ull;
    const uint32_t base = (uint32_t)(start - a0), n16 = (base + nkp * stride + 15) >> 4;
    for (uint32_t i = tid; i < n16; i += kWireBlk) sraw[i] = stage[(a0 >> 4) + i];
    __syncthreads();
    const uint32_t* __restrict__ l32 = reinterpret_cast<const uint32_t*>(sraw);
    uint32_t* __restrict__ desc = reinterpret_cast<uint32_t*>(arena + sg.desc_off) + (size_t)kp0 * W;
    for (uint32_t item = tid; item < nkp * W; item += kWireBlk) {
        const uint32_t i = item / W, w = item - i * W;
        const uint32_t rec = base + i * stride, d0 = rec + 17 + 16 * w;
        const uint32_t a = d0 >> 2, sh = (d0 & 3) * 8;
        const uint32_t x0 = l32[a], x1 = l32[a + 1], x2 = l32[a + 2], x3 = l32[a + 3], x4 = sh ? l32[a + 4] : 0u;
        desc[item] = float_to_byte(__funnelshift_r(x0, x1, sh)) | float_to_byte(__funnelshift_r(x1, x2, sh)) << 8 |
                     float_to_byte(__funnelshift_r(x2, x3, sh)) << 16 | float_to_byte(__funnelshift_r(x3, x4, sh)) << 24;
        const uint32_t p0 = rec + 17 + 16 * W;
        uint32_t* pos = reinterpret_cast<uint32_t*>(arena + sg.pos_off) + (size_t)(kp0 + i) * 6;          // column of the 3 x n matrix
        for (uint32_t k = w; k < 6; k += W) pos[k] = lds_u32_at(l32, p0 + 4 * k);
        if (w == W - 1) {
            if (lds_u32_at(l32, rec + 13) != 4 * W) atomicOr(bad, 1);
            arena[sg.valid_off + kp0 + i] = (lds_u32_at(l32, rec + 8) & 0xffu) ? 1 : 0;                   // std::vector<bool>::push_back(is_3d) (:164)
        }
        if (uv && w == (W > 1 ? W - 2 : 0)) {
            uv[2 * (sg.feat_begin + kp0 + i)] = (int32_t)lds_u32_at(l32, rec);
            uv[2 * (sg.feat_begin + kp0 + i) + 1] = (int32_t)lds_u32_at(l32, rec + 4);
        }
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

// keypoints per workgroup for a record stride: as many as fit the LDS stage, at most 128
int wire_kpb(uint32_t stride) { return (int)std::max<uint32_t>(1u, std::min<uint32_t>(128u, (kWireLdsBytes - 16) / stride)); }

void launch_wire_unpack(const uint32_t* stage, uint8_t* arena, const WireSeg* segs, int n_segs, int64_t n_blocks, int32_t* uv, int32_t* bad,
                        hipStream_t s)
{
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(wire_unpack_kernel, dim3((unsigned)n_blocks), dim3(kWireBlk), 0, s, reinterpret_cast<const uint4*>(stage), arena, segs, n_segs,
                       uv, bad);
}

void launch_wire_pack(const uint8_t* arena, const WireSeg& sg, const int32_t* uv, uint32_t* out, uint64_t n_bytes, hipStream_t s)
{
    const int64_t n_dwords = (int64_t)((n_bytes + 3) / 4);
    if (n_dwords <= 0) return;
    const int64_t blocks = (n_dwords + kWireBlk - 1) / kWireBlk;
    hipLaunchKernelGGL(wire_pack_kernel, dim3((unsigned)blocks), dim3(kWireBlk), 0, s, arena, sg, uv, out, n_dwords, n_bytes);
}

}  // namespace uzl
