// grok_amd/csrc/kernels_ingest.hip -- K1: widen + DC level shift + RCT/ICT, gfx950.
//
// Replaces TileProcessor::copy_uncompressed_data_to_tile (tile/TileProcessor.cpp:1166-1216),
// dc_level_shift_encode (:922-944) and mct::compress_rev / compress_irrev
// (point_transform/mct.cpp:48-105, :469-554) with one HBM-bound pass:
// reads b_in bytes/sample, writes 4 bytes/sample (S*(b_in+4) algorithmic bytes).
// Each lane handles 4 consecutive samples of a row: one 4/8-byte load per component and one
// 16-byte store per component, so a wave moves 256..512 B in and 1 KiB out per instruction.
#include "kernels.h"

namespace grk_amd {

__device__ __forceinline__ void color_fwd(int32_t& c0, int32_t& c1, int32_t& c2, bool irrev)
{
    if (!irrev) {
        // RCT (mct.cpp:94-104)
        int32_t r = c0, g = c1, b = c2;
        c0 = (r + 2 * g + b) >> 2;
        c1 = b - g;
        c2 = r - g;
    } else {
        // ICT (mct.cpp:541-553): every product/sum rounded separately, left-to-right adds.
        const float a_r = 0.299f, a_g = 0.587f, a_b = 0.114f;
        const float cb = 0.5f / (1.0f - a_b), cr = 0.5f / (1.0f - a_r);
        float r = (float)c0, g = (float)c1, b = (float)c2;
        float y = __fmul_rn(a_r, r);
        y = __fadd_rn(y, __fmul_rn(a_g, g));
        y = __fadd_rn(y, __fmul_rn(a_b, b));
        float u = __fmul_rn(cb, __fsub_rn(b, y));
        float v = __fmul_rn(cr, __fsub_rn(r, y));
        c0 = __float_as_int(y); c1 = __float_as_int(u); c2 = __float_as_int(v);
    }
}

template <typename PIX>
__device__ __forceinline__ void load4(const PIX* p, bool vec, uint32_t n, int32_t v[4])
{
    if (vec) {
        if constexpr (sizeof(PIX) == 1) {
            uchar4 q = *reinterpret_cast<const uchar4*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            ushort4 q = *reinterpret_cast<const ushort4*>(p);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
    } else {
        for (uint32_t i = 0; i < 4; ++i) v[i] = i < n ? (int32_t)p[i] : 0;
    }
}

template <typename PIX, int NC>
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a)
{
    const uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 4u;
    const uint32_t y = blockIdx.y;
    const uint32_t tile = blockIdx.z;
    if (x >= a.w) return;
    const uint32_t n = a.w - x < 4 ? a.w - x : 4;
    const size_t comp_px = (size_t)a.w * a.h;
    const PIX* src = reinterpret_cast<const PIX*>(a.pixels) + (size_t)tile * a.ncomp * comp_px + (size_t)y * a.w + x;
    int32_t* dst = a.planes + (size_t)tile * a.ncomp * a.pitch + (size_t)y * a.stride + x;
    // rows are tightly packed: 4-sample vectors are aligned only when w % 4 == 0
    const bool vec = (n == 4) && ((a.w & 3u) == 0);
    const bool irrev = a.irreversible != 0;

    // NC is a compile-time constant so that c[][] lives in registers (a runtime bound spills to scratch)
    int32_t c[NC][4];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        load4<PIX>(src + (size_t)k * comp_px, vec, n, c[k]);
        for (int i = 0; i < 4; ++i) c[k][i] = ((c[k][i] ^ a.sext) - a.sext) - a.dc;     // sign-extend int8/int16, DC shift
    }
    if (NC >= 3 && a.mct) {
        for (int i = 0; i < 4; ++i) color_fwd(c[0][i], c[NC >= 3 ? 1 : 0][i], c[NC >= 3 ? 2 : 0][i], irrev);
    } else if (irrev) {
        // 9/7 without MCT: the transform works on floats
#pragma unroll
        for (int k = 0; k < NC; ++k)
            for (int i = 0; i < 4; ++i) c[k][i] = __float_as_int((float)c[k][i]);
    }
    if (NC > 3 && a.mct && irrev)
        for (int i = 0; i < 4; ++i) c[NC > 3 ? 3 : 0][i] = __float_as_int((float)c[NC > 3 ? 3 : 0][i]);
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        int32_t* d = dst + (size_t)k * a.pitch;
        if (n == 4) {
            *reinterpret_cast<int4*>(d) = make_int4(c[k][0], c[k][1], c[k][2], c[k][3]);   // stride % 32 == 0, x % 4 == 0
        } else {
            for (uint32_t i = 0; i < n; ++i) d[i] = c[k][i];
        }
    }
}

hipError_t launch_ingest(const IngestArgs& a, hipStream_t s)
{
    dim3 grid((a.w + 1023) / 1024, a.h, a.ntiles), block(256);
#define GRK_INGEST(PIX)                                                                          \
    switch (a.ncomp) {                                                                          \
    case 1: hipLaunchKernelGGL((ingest_kernel<PIX, 1>), grid, block, 0, s, a); break;           \
    case 2: hipLaunchKernelGGL((ingest_kernel<PIX, 2>), grid, block, 0, s, a); break;           \
    case 3: hipLaunchKernelGGL((ingest_kernel<PIX, 3>), grid, block, 0, s, a); break;           \
    default: hipLaunchKernelGGL((ingest_kernel<PIX, 4>), grid, block, 0, s, a); break;          \
    }
    if (a.bytes_per_sample == 1) { GRK_INGEST(uint8_t) } else { GRK_INGEST(uint16_t) }
#undef GRK_INGEST
    return hipGetLastError();
}

// ---- per-block energy of the quantised coefficients: the rate-control hook ----------------------------------
// One wave per code-block of the latest encode: sum over the block's samples of q^2, q the integer magnitude K3 codes -- |x| for the
// reversible path, trunc(|c| * (1 / step)) for the irreversible one (the quantiser of kernels_ht.hip, T1HT.cpp:58-101 as intended)
// -- exact in 64 bits (q < 2^26, 4096 samples).  The host turns it into grk_plugin_pass::distortionDecrease with the weights the
// reference's Tier-1 uses for its own passes (T1::getwmsedec, t1/t1_part1/T1.cpp:394-414).
template <class PLANE, bool IRREV>
__global__ __launch_bounds__(64) void block_energy_kernel(const PLANE* mallat, uint32_t stride, uint64_t pitch, const HtBlockDesc* blocks,
                                                          uint32_t blocks_per_tile, uint32_t ncomp, unsigned long long* out)
{
    const uint32_t i = blockIdx.x, lane = threadIdx.x;
    const HtBlockDesc bd = blocks[i % blocks_per_tile];
    const uint32_t tile = i / blocks_per_tile;
    const PLANE* src = mallat + ((size_t)tile * ncomp + bd.comp) * pitch + (size_t)bd.py * stride + bd.px;
    unsigned long long acc = 0;
    const uint32_t lim = (1u << bd.kmax) - 1u;
    for (uint32_t y = 0; y < bd.h; ++y)
        for (uint32_t x = lane; x < bd.w; x += 64) {
            uint32_t q;
            if constexpr (IRREV) {
                const float c = __int_as_float((int32_t)src[(size_t)y * stride + x]);
                q = (uint32_t)__fmul_rn(fabsf(c), bd.inv_step);
                q = q > lim ? lim : q;
            } else {
                const int32_t v = (int32_t)src[(size_t)y * stride + x];
                q = (uint32_t)(v < 0 ? -v : v);
            }
            acc += (unsigned long long)q * q;
        }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) out[i] = acc;
}

hipError_t launch_block_energy(const void* mallat, int h16, int irreversible, uint32_t stride, uint64_t pitch, const HtBlockDesc* blocks,
                               uint32_t blocks_per_tile, uint32_t ncomp, uint64_t nblocks, unsigned long long* out, hipStream_t s)
{
    if (!nblocks) return hipSuccess;
    const dim3 grid((uint32_t)nblocks), block(64);
    if (h16) hipLaunchKernelGGL((block_energy_kernel<int16_t, false>), grid, block, 0, s, (const int16_t*)mallat, stride, pitch, blocks, blocks_per_tile, ncomp, out);
    else if (irreversible) hipLaunchKernelGGL((block_energy_kernel<int32_t, true>), grid, block, 0, s, (const int32_t*)mallat, stride, pitch, blocks, blocks_per_tile, ncomp, out);
    else hipLaunchKernelGGL((block_energy_kernel<int32_t, false>), grid, block, 0, s, (const int32_t*)mallat, stride, pitch, blocks, blocks_per_tile, ncomp, out);
    return hipGetLastError();
}

} // namespace grk_amd
