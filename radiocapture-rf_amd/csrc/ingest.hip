// ingest.hip -- on-device conversion of the SDR wire formats to cf32.
//
// The reference receives cf32 because gr-osmosdr / gr-uhd convert the hardware's samples on the host
// (/root/reference/rc_frontend/receiver.py:74-98,170-191: rtl-sdr delivers unsigned 8-bit I/Q, USRP /
// bladeRF signed 16-bit).  Converting here instead moves 2 or 4 bytes per sample over PCIe rather
// than 8 -- the host link (63 GB/s), not HBM, is the real end-to-end bottleneck of this path.
//   x = (float(raw) - offset) * scale     (float32, unfused: the drivers' lookup-table semantics)
// Bound: HBM, 2|4 B read + 8 B written per sample; 16 bytes stored per lane.
#include <algorithm>
#include <cstdlib>

#include "rcf_internal.h"

namespace rcfx {

namespace {

template <typename T>
__device__ __forceinline__ float conv1(T r, float scale, float offset)
{
    return __fmul_rn(__fsub_rn((float)r, offset), scale);
}

// one thread = two complex samples = four raw values = one 16-byte store (grid-stride)
template <typename T>
__global__ __launch_bounds__(256) void convert_kernel(const T *__restrict__ raw, float4 *__restrict__ out,
                                                      size_t n_pairs, float scale, float offset)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n_pairs; i += stride) {
        const T *r = raw + 4 * i;
        out[i] = make_float4(conv1(r[0], scale, offset), conv1(r[1], scale, offset), conv1(r[2], scale, offset),
                             conv1(r[3], scale, offset));
    }
}

template <typename T>
__global__ void convert_tail(const T *__restrict__ raw, float2 *__restrict__ out, float scale, float offset)
{
    if (threadIdx.x == 0) *out = make_float2(conv1(raw[0], scale, offset), conv1(raw[1], scale, offset));
}

template <typename T>
void launch_t(const void *raw, float2 *out, size_t n, float scale, float offset, hipStream_t s)
{
    const T *r = static_cast<const T *>(raw);
    const size_t n_pairs = n / 2;
    if (n_pairs) {
        const int grid = (int)std::min<size_t>((n_pairs + 255) / 256, 256 * 8);
        hipLaunchKernelGGL((convert_kernel<T>), dim3(grid), dim3(256), 0, s, r, reinterpret_cast<float4 *>(out),
                           n_pairs, scale, offset);
    }
    if (n & 1)
        hipLaunchKernelGGL((convert_tail<T>), dim3(1), dim3(64), 0, s, r + 2 * (n - 1), out + (n - 1), scale, offset);
}

__global__ __launch_bounds__(256) void gather_view_kernel(StreamView v, int64_t first, float2 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = v.base[v.at(first + (int64_t)i)];
}

// plain 8-byte-granular copy.  The per-block schedule used hipMemcpyAsync for its two small copies (launch records
// host -> device, history tail device -> device); the runtime puts each of those behind ~6 us of queue gap, while a
// kernel follows the kernel before it with none (rocprof trace of the timed configuration: 136.8 us per step of which
// 12 us were those two gaps).  n8 = number of 8-byte words; src may be pinned host memory.
__global__ __launch_bounds__(256) void copy8_kernel(unsigned long long *__restrict__ dst,
                                                    const unsigned long long *__restrict__ src, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

}  // namespace

// two such copies in one launch (the block's launch records and its history tail: one launch less per block)
static __global__ __launch_bounds__(256) void copy8x2_kernel(unsigned long long *__restrict__ d0, const unsigned long long *__restrict__ s0,
                                                      size_t n0, unsigned long long *__restrict__ d1,
                                                      const unsigned long long *__restrict__ s1, size_t n1)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += stride) {
        if (i < n0) d0[i] = s0[i];
        else d1[i - n0] = s1[i - n0];
    }
}

void launch_copy8x2(void *d0, const void *s0, size_t bytes0, void *d1, const void *s1, size_t bytes1, hipStream_t s)
{
    const size_t n0 = (bytes0 + 7) / 8, n1 = (bytes1 + 7) / 8;
    if (n0 + n1 == 0) return;
    const size_t blocks = std::min<size_t>((n0 + n1 + 255) / 256, 2048);
    hipLaunchKernelGGL(copy8x2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<unsigned long long *>(d0),
                       static_cast<const unsigned long long *>(s0), n0, static_cast<unsigned long long *>(d1),
                       static_cast<const unsigned long long *>(s1), n1);
}

void launch_copy8(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    const size_t n8 = (bytes + 7) / 8;
    if (n8 == 0) return;
    const size_t blocks = std::min<size_t>((n8 + 255) / 256, 2048);
    hipLaunchKernelGGL(copy8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<unsigned long long *>(dst),
                       static_cast<const unsigned long long *>(src), n8);
}

// rcf_chan_read_many: the new samples of many channel rings packed back to back into ONE staging buffer (pinned host
// memory the device writes across PCIe) -- one launch and one synchronisation per egress pass instead of a device
// round trip per channel.  Records live in pinned host memory too.  Units: 4-byte words.
// A BOUNDED grid, like group_prep_kernel's and for the same reason: its stores cross PCIe, a workgroup whose stores wait for the
// link holds its CU slot, and launched one workgroup per (record, part) -- tens of thousands for a group's read -- it
// starved the filterbank launches of the other groups (rocprof of the real-time leg: a 20 us pfb5 launch averaging 183 us).
static __global__ __launch_bounds__(256) void gather_rings_kernel(const GatherRec *__restrict__ recs, uint32_t *__restrict__ dst,
                                                                  uint32_t n_recs, uint32_t parts)
{
    const uint32_t n_items = n_recs * parts;
    for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
        const GatherRec r = recs[item / parts];
        const uint32_t stride = parts * 256;
        // the destination is either linear (dst_mask_w = ~0: rows packed back to back, rcf_chan_read_many) or a ring of its
        // own (the real-time pump's per-channel host rings: dst_w = the ring's first word, dst_pos_w where this segment starts)
        for (uint32_t w = (item % parts) * 256 + threadIdx.x; w < r.n_w; w += stride) {
            const uint32_t i = (r.pos_w + w) & r.mask_w;
            uint32_t v = r.ring[r.stride_w > 1 ? (size_t)i * r.stride_w : (size_t)i];
            if (r.flags & 1u) v = __float_as_uint(__fmul_rn(r.gain, __uint_as_float(v)));
            dst[r.dst_w + ((r.dst_pos_w + w) & r.dst_mask_w)] = v;
        }
    }
}

void launch_gather_rings(const GatherRec *d_recs, int n_recs, uint32_t *d_dst, uint32_t max_words, hipStream_t s)
{
    if (n_recs <= 0 || max_words == 0) return;
    static const unsigned cap = [] { const char *e = getenv("RCF_GATHER_WGS"); const int v = e ? atoi(e) : 0; return (unsigned)(v > 0 ? v : 128); }();
    const unsigned parts = std::min<unsigned>((max_words + 255) / 256, 16);
    const unsigned items = parts * (unsigned)n_recs;
    hipLaunchKernelGGL(gather_rings_kernel, dim3(std::min(items, cap)), dim3(256), 0, s, d_recs, d_dst, (uint32_t)n_recs, parts);
}

// one bin's discriminator samples out of a frame-major ring of floats (rcf_pfb_read_fm): dst[i] = gain * base[((first + i) & mask) stride]
static __global__ void gather_f32_kernel(const float *__restrict__ base, uint64_t mask, int64_t stride, int64_t first, float gain,
                                         float *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = __fmul_rn(gain, base[((uint64_t)(first + (int64_t)i) & mask) * (uint64_t)stride]);
}

void launch_gather_f32(const float *base, uint64_t mask, int64_t stride, int64_t first, float gain, float *dst, size_t n, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, base, mask, stride, first, gain, dst, n);
}

void launch_gather_view(const StreamView &v, int64_t first, float2 *dst, size_t n, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(gather_view_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, first, dst, n);
}

// ---------------------------------------------------------------- grouped ingest (rcf_group.cpp)
// ONE launch for the blocks of G front-ends: wire-format (or cf32) samples -> each front-end's wideband buffer, the
// history tail of every block behind the OTHER buffer of its front-end (dual-written on the way, so no copy launch
// follows), and plain 8-byte copies (the group's launch records host -> device arena; the part of a history that is
// older than a short block).  grid = (tiles, records); the records themselves are read where the host wrote them
// (pinned, device-mapped arena): 64 bytes once per workgroup.
namespace {

constexpr int kPrepTile = 2048;      // samples (or 8-byte words) per tile: 256 threads x 2 x 4

template <typename T, int NV>
__device__ __forceinline__ void prep_load(const PrepRec &r, uint32_t i, float (&v)[4])
{
    // two complex samples = four raw values starting at sample i (i even, i + 1 < n)
    const T *p = static_cast<const T *>(r.src) + 2 * (size_t)i;
    T raw[4];
    if (r.aligned) {
        typedef T vec_t __attribute__((ext_vector_type(4)));
        const vec_t q = *reinterpret_cast<const vec_t *>(p);            // ONE 4 / 8-byte load per lane
        raw[0] = q.x; raw[1] = q.y; raw[2] = q.z; raw[3] = q.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = p[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = conv1(raw[u], r.scale, r.offset);
}

__device__ __forceinline__ void prep_store(const PrepRec &r, uint32_t i, int cnt, const float (&v)[4])
{
    float2 *d = r.dst + i;
    if (cnt == 2 && r.dst_aligned) *reinterpret_cast<float4 *>(d) = make_float4(v[0], v[1], v[2], v[3]);
    else { d[0] = make_float2(v[0], v[1]); if (cnt == 2) d[1] = make_float2(v[2], v[3]); }
    if (r.hist_dst) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (u < cnt && i + u >= r.hist_from) r.hist_dst[i + u - r.hist_from] = make_float2(v[2 * u], v[2 * u + 1]);
    }
}

// one tile (kPrepTile samples or words from `base` on) of one record
__device__ __forceinline__ void prep_tile(const PrepRec &r, const uint32_t base, const int tid)
{
    if (base >= r.n) return;
    if (r.fmt < 0) {                                   // plain copy, n 8-byte words
        const unsigned long long *s = static_cast<const unsigned long long *>(r.src);
        unsigned long long *d = reinterpret_cast<unsigned long long *>(r.dst);
        unsigned long long w[kPrepTile / 256];
#pragma unroll
        for (int u = 0; u < kPrepTile / 256; ++u) { const uint32_t i = base + u * 256 + tid; w[u] = i < r.n ? s[i] : 0ull; }
#pragma unroll
        for (int u = 0; u < kPrepTile / 256; ++u) { const uint32_t i = base + u * 256 + tid; if (i < r.n) d[i] = w[u]; }
        return;
    }
    float v[4][4];
    int cnt[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                      // all four loads of the lane before the first store
        const uint32_t i = base + 2 * (u * 256 + tid);
        cnt[u] = i + 1 < r.n ? 2 : (i < r.n ? 1 : 0);
        if (cnt[u] == 2) {
            switch (r.fmt) {
                case RCF_FMT_U8:  prep_load<uint8_t, 4>(r, i, v[u]); break;
                case RCF_FMT_S8:  prep_load<int8_t, 4>(r, i, v[u]); break;
                case RCF_FMT_S16: prep_load<int16_t, 4>(r, i, v[u]); break;
                default: {
                    const float *p = static_cast<const float *>(r.src) + 2 * (size_t)i;
                    if (r.aligned) { const float4 q = *reinterpret_cast<const float4 *>(p); v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w; }
                    else { v[u][0] = p[0]; v[u][1] = p[1]; v[u][2] = p[2]; v[u][3] = p[3]; }
                }
            }
        } else if (cnt[u] == 1) {                      // the odd last sample of a block
            v[u][2] = v[u][3] = 0.f;
            switch (r.fmt) {
                case RCF_FMT_U8:  { const uint8_t *p = static_cast<const uint8_t *>(r.src) + 2 * (size_t)i; v[u][0] = conv1(p[0], r.scale, r.offset); v[u][1] = conv1(p[1], r.scale, r.offset); break; }
                case RCF_FMT_S8:  { const int8_t *p = static_cast<const int8_t *>(r.src) + 2 * (size_t)i; v[u][0] = conv1(p[0], r.scale, r.offset); v[u][1] = conv1(p[1], r.scale, r.offset); break; }
                case RCF_FMT_S16: { const int16_t *p = static_cast<const int16_t *>(r.src) + 2 * (size_t)i; v[u][0] = conv1(p[0], r.scale, r.offset); v[u][1] = conv1(p[1], r.scale, r.offset); break; }
                default:          { const float *p = static_cast<const float *>(r.src) + 2 * (size_t)i; v[u][0] = p[0]; v[u][1] = p[1]; }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (cnt[u]) prep_store(r, base + 2 * (u * 256 + tid), cnt[u], v[u]);
}

// A BOUNDED grid: the kernel is PCIe-bound (its loads cross the link), and a workgroup that waits for the link holds its
// slot on a CU -- launched one workgroup per tile (thousands), the blocks of one group's ingest starved the OTHER groups'
// filterbank launches of CUs (rocprof of the real-time leg: 27 front-ends' filterbank 132 us instead of ~40).  So
// 128 workgroups (RCF_PREP_WGS) each walk a contiguous range of the launch's tiles; records[i].tile_first (host) says
// where record i's tiles start, all of them are fetched into LDS once, the 64-byte record itself only when the range
// crosses into the next record.
__global__ __launch_bounds__(256) void group_prep_kernel(const PrepRec *__restrict__ recs, int n_recs, uint32_t total_tiles)
{
    __shared__ uint32_t tf[kPrepMaxRecs + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < n_recs; i += 256) tf[i] = recs[i].tile_first;
    if (tid == 0) tf[n_recs] = total_tiles;
    __syncthreads();
    const uint32_t per = (total_tiles + gridDim.x - 1) / gridDim.x;
    uint32_t t = blockIdx.x * per;
    const uint32_t t_end = min(total_tiles, t + per);
    if (t >= t_end) return;
    int lo = 0, hi = n_recs;                           // record of tile t: tf[lo] <= t < tf[lo + 1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (tf[mid] <= t) lo = mid; else hi = mid;
    }
    int ri = __builtin_amdgcn_readfirstlane(lo);
    PrepRec r = recs[ri];
    for (; t < t_end; ++t) {
        while (t >= tf[ri + 1]) { ++ri; r = recs[ri]; }    // (records without tiles -- n = 0 -- are stepped over)
        prep_tile(r, (t - tf[ri]) * (uint32_t)kPrepTile, tid);
    }
}

}  // namespace

// recs: as the DEVICE reads them (the pinned arena); tile_first filled in by fill_prep_tiles().  At most kPrepMaxRecs per launch.
void launch_group_prep(const PrepRec *d_recs, int n_recs, uint32_t total_tiles, hipStream_t s)
{
    if (n_recs <= 0 || total_tiles == 0) return;
    static const unsigned cap = [] { const char *e = getenv("RCF_PREP_WGS"); const int v = e ? atoi(e) : 0; return (unsigned)(v > 0 ? v : 128); }();   // 128: same PCIe rate as 512 (0.6 ms per 27 blocks), the filterbank launches beside it 96 -> 68 us
    hipLaunchKernelGGL(group_prep_kernel, dim3(std::min<unsigned>(cap, total_tiles)), dim3(256), 0, s, d_recs, n_recs, total_tiles);
}

uint32_t fill_prep_tiles(PrepRec *recs, int n_recs)
{
    uint32_t t = 0;
    for (int i = 0; i < n_recs; ++i) {
        recs[i].tile_first = t;
        t += (recs[i].n + (uint32_t)kPrepTile - 1) / (uint32_t)kPrepTile;
    }
    return t;
}

size_t raw_sample_bytes(int fmt)
{
    switch (fmt) {
        case RCF_FMT_U8:
        case RCF_FMT_S8:  return 2;
        case RCF_FMT_S16: return 4;
        default:          return 0;
    }
}

void launch_convert(int fmt, const void *d_raw, float2 *d_out, size_t n, float scale, float offset, hipStream_t s)
{
    switch (fmt) {
        case RCF_FMT_U8:  launch_t<uint8_t>(d_raw, d_out, n, scale, offset, s); break;
        case RCF_FMT_S8:  launch_t<int8_t>(d_raw, d_out, n, scale, offset, s); break;
        case RCF_FMT_S16: launch_t<int16_t>(d_raw, d_out, n, scale, offset, s); break;
        default: break;
    }
}

}  // namespace rcfx
