// grok_amd/csrc/kernels_ht.hip -- K3: HTJ2K cleanup-pass encoder, one wavefront per code-block.
//
// Replaces T1HT::preCompress + ojph_encode_codeblock (t1/t1_ht/T1HT.cpp:58-128,
// t1/t1_ht/coding/ojph_block_encoder.cpp:463-938), which the reference runs one block per CPU
// thread and strictly serially inside the block.
//
// Wave64 decomposition (lane <-> quad, two quad rows of 32 quads per iteration):
//   phase A (all 64 lanes): sample -> (rho, exponents, MagSgn values) per quad; the context of a
//       quad only needs its left neighbour's rho and the exponent/significance of the sample row
//       above, both fetched with __shfl; VLC tuple lookup; per-quad MagSgn bit count and
//       per-quad-pair VLC+UVLC bit count; wave prefix sums give every codeword its bit offset in
//       the raw (un-stuffed) MagSgn / VLC streams, which are assembled in LDS with ds_or;
//       MEL events are gathered with __ballot and run through the 13-state MEL coder in
//       wave-uniform (scalar) code.
//   phase B: byte-stuffing + termination + concatenation MagSgn | MEL | VLC(reversed) + Scup.
//
// Bit-exactness contract: the byte string per block equals ojph_encode_codeblock's (tests compare
// against oracle/ and the reference build).
#include "kernels.h"
#include "ht_vlc_tables.h"

namespace grk_amd {

namespace {

__device__ __constant__ uint8_t kMelE[13] = {0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 4, 5};

// device copies of the generated encoder tables: [0..2047] first quad row, [2048..4095] others
__device__ uint16_t g_vlc_enc[4096];

struct MelState {
    int run, k, acc, left;
    uint32_t pos;
};

__device__ __forceinline__ void mel_put_bit(MelState& m, uint8_t* buf, int v, bool writer)
{
    m.acc = (m.acc << 1) | v;
    if (--m.left == 0) {
        if (writer) buf[m.pos] = (uint8_t)m.acc;
        m.pos++;
        m.left = (m.acc == 0xFF) ? 7 : 8;
        m.acc = 0;
    }
}
__device__ __forceinline__ void mel_event(MelState& m, uint8_t* buf, int one, bool writer)
{
    const int e = kMelE[m.k];
    if (!one) {
        if (++m.run >= (1 << e)) {
            mel_put_bit(m, buf, 1, writer);
            m.run = 0;
            if (m.k < 12) m.k++;
        }
    } else {
        mel_put_bit(m, buf, 0, writer);
        for (int t = e; t > 0;) mel_put_bit(m, buf, (m.run >> --t) & 1, writer);
        m.run = 0;
        if (m.k > 0) m.k--;
    }
}

__device__ __forceinline__ void or_bits(uint32_t* raw, uint32_t pos, uint32_t val, uint32_t n)
{
    if (n == 0) return;
    const uint32_t w = pos >> 5, sh = pos & 31;
    atomicOr(&raw[w], val << sh);
    if (sh + n > 32) atomicOr(&raw[w + 1], val >> (32 - sh));
}
__device__ __forceinline__ void or_bits64(uint32_t* raw, uint32_t pos, uint64_t val, uint32_t n)
{
    if (n == 0) return;
    const uint32_t w = pos >> 5, sh = pos & 31;
    atomicOr(&raw[w], (uint32_t)(val << sh));
    if (sh + n > 32) {
        const uint64_t hi = val >> (32 - sh);
        atomicOr(&raw[w + 1], (uint32_t)hi);
        if (sh + n > 64) atomicOr(&raw[w + 2], (uint32_t)(hi >> 32));
    }
}
// n <= 25 bits starting at bit `pos` of a little-endian bit array
__device__ __forceinline__ uint32_t get_bits(const uint32_t* raw, uint32_t pos, uint32_t n)
{
    const uint32_t w = pos >> 5, sh = pos & 31;
    uint64_t v = raw[w] | ((uint64_t)raw[w + 1] << 32);
    return (uint32_t)(v >> sh) & ((1u << n) - 1);
}

__device__ __forceinline__ void uvlc(int u, uint32_t& pre, uint32_t& pl, uint32_t& suf, uint32_t& sl)
{   // ojph_block_encoder.cpp:189-210
    if (u <= 2)      { pre = (uint32_t)u; pl = (uint32_t)u; suf = 0; sl = 0; }
    else if (u <= 4) { pre = 4; pl = 3; suf = (uint32_t)(u - 3); sl = 1; }
    else             { pre = 0; pl = 3; suf = (uint32_t)(u - 5); sl = 5; }
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

template <bool IRREV>
__global__ __launch_bounds__(64) void ht_encode_kernel(HtArgs a, uint32_t ms_words, uint32_t vlc_words)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* ms_raw  = smem;
    uint32_t* vlc_raw = smem + ms_words;
    uint8_t*  mel_buf = reinterpret_cast<uint8_t*>(smem + ms_words + vlc_words);

    const int lane = threadIdx.x;
    const uint32_t gid = blockIdx.x;
    const uint32_t tile = gid / a.blocks_per_tile;
    const HtBlockDesc bd = a.blocks[gid % a.blocks_per_tile];
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    const int32_t* src = a.mallat + ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    const uint32_t p = 30u - bd.kmax;

    for (uint32_t i = lane; i < ms_words + vlc_words + 64; i += 64) smem[i] = 0;
    __syncthreads();
    if (lane == 0) vlc_raw[0] = 0xF;             // vlc_init: four 1 bits pending (:315-318)
    __syncthreads();

    MelState mel{0, 0, 0, 8, 0};
    uint32_t ms_bits = 0, vlc_bits = 4;
    uint32_t Bprev = 0;
    const uint32_t qx = lane & 31, half = lane >> 5;
    const uint32_t iters = (QH + 1) >> 1;

    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t qy = 2 * it + half;
        const bool active = qx < QW && qy < QH;
        // ---- samples -> sign-magnitude words (T1HT.cpp:71-84 / dead-zone quantiser) ----------
        uint32_t tw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = 2 * qx + (i >> 1), y = 2 * qy + (i & 1);
            uint32_t t = 0;
            if (x < w && y < h) {
                const int32_t raw = src[(size_t)y * a.stride + x];
                if constexpr (IRREV) {
                    const float c = __int_as_float(raw);
                    float q = __fmul_rn(fabsf(c), bd.inv_step);
                    uint32_t mag = (uint32_t)q;
                    const uint32_t lim = (1u << bd.kmax) - 1u;
                    mag = mag > lim ? lim : mag;
                    t = ((c < 0.f && mag) ? 0x80000000u : 0u) | (mag << p);
                } else {
                    const uint32_t mag = (uint32_t)(raw < 0 ? -raw : raw);
                    t = (raw < 0 ? 0x80000000u : 0u) | (mag << p);
                }
            }
            tw[i] = t;
        }
        // ---- per-sample analysis (:513-563) ---------------------------------------------------
        uint32_t rho = 0, emax = 0, e[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t val = ((tw[i] + tw[i]) >> p) & ~1u;
            e[i] = 0; v[i] = 0;
            if (val) {
                rho |= 1u << i;
                e[i] = 32u - (uint32_t)__clz((int)(val - 1));
                emax = max(emax, e[i]);
                v[i] = val - 2 + (tw[i] >> 31);
            }
        }
        // ---- neighbourhood ---------------------------------------------------------------------
        const uint32_t Bcur = e[1] | (e[3] << 8) | (((rho >> 1) & 1) << 16) | (((rho >> 3) & 1) << 17);
        const uint32_t sel = half ? Bprev : Bcur;
        const uint32_t above = __shfl_xor(sel, 32);
        uint32_t above_l = __shfl_up(above, 1);   if (qx == 0)  above_l = 0;
        uint32_t above_r = __shfl_down(above, 1); if (qx == 31) above_r = 0;
        uint32_t rho_left = __shfl_up(rho, 1);    if (qx == 0)  rho_left = 0;
        uint32_t c_q, kappa;
        if (qy == 0) {
            c_q = (rho_left >> 1) | (rho_left & 1);
            kappa = 1;
        } else {
            const uint32_t e_w = (above_l >> 8) & 0xFF, e_n0 = above & 0xFF, e_n1 = (above >> 8) & 0xFF, e_e = above_r & 0xFF;
            const int max_e = (int)max(max(e_w, e_n0), max(e_n1, e_e)) - 1;
            const uint32_t s_w = ((above_l >> 17) | (above >> 16)) & 1, s_e = ((above >> 17) | (above_r >> 16)) & 1;
            c_q = s_w | ((((rho_left >> 2) | (rho_left >> 3)) & 1) << 1) | (s_e << 2);
            kappa = (rho & (rho - 1)) ? (uint32_t)max(1, max_e) : 1u;
        }
        const uint32_t U = max(emax, kappa);
        const uint32_t u = U - kappa;
        uint32_t eps = 0;
        if (u > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) eps |= (uint32_t)(e[i] == emax) << i;
        }
        uint32_t tuple = 0;
        if (active) tuple = g_vlc_enc[(qy == 0 ? 0u : 2048u) + ((c_q << 8) | (rho << 4) | eps)];
        // ---- MagSgn: bit counts, offsets, emission --------------------------------------------
        uint32_t m[4], ms_len = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = ((rho >> i) & 1) ? U - ((tuple >> i) & 1) : 0;
            ms_len += m[i];
        }
        const uint32_t ms_incl = wave_incl_scan(ms_len, lane);
        uint32_t mpos = ms_bits + ms_incl - ms_len;
        ms_bits += __shfl(ms_incl, 63);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (m[i]) {
                const uint32_t mask = m[i] >= 32 ? 0xFFFFFFFFu : ((1u << m[i]) - 1u);
                or_bits(ms_raw, mpos, v[i] & mask, m[i]);
                mpos += m[i];
            }
        }
        // ---- VLC + UVLC per quad pair (even lane assembles) ------------------------------------
        const uint32_t tuple_p = __shfl_xor(tuple, 1);
        const uint32_t u_p = __shfl_xor(u, 1);
        uint64_t cw = 0; uint32_t cl = 0;
        if ((qx & 1) == 0 && active) {
            const uint32_t u0 = u, u1 = u_p;                  // partner inactive -> tuple_p = 0, u1 = 0
            cw = tuple >> 8; cl = (tuple >> 4) & 7;
            cw |= (uint64_t)(tuple_p >> 8) << cl; cl += (tuple_p >> 4) & 7;
            uint32_t p0, l0, s0, sl0, p1, l1, s1, sl1;
            if (qy == 0 && u0 > 2 && u1 > 2) {
                uvlc((int)u0 - 2, p0, l0, s0, sl0); uvlc((int)u1 - 2, p1, l1, s1, sl1);
                cw |= (uint64_t)p0 << cl; cl += l0; cw |= (uint64_t)p1 << cl; cl += l1;
                cw |= (uint64_t)s0 << cl; cl += sl0; cw |= (uint64_t)s1 << cl; cl += sl1;
            } else if (qy == 0 && u0 > 2 && u1 > 0) {
                uvlc((int)u0, p0, l0, s0, sl0);
                cw |= (uint64_t)p0 << cl; cl += l0; cw |= (uint64_t)(u1 - 1) << cl; cl += 1;
                cw |= (uint64_t)s0 << cl; cl += sl0;
            } else {
                uvlc((int)u0, p0, l0, s0, sl0); uvlc((int)u1, p1, l1, s1, sl1);
                cw |= (uint64_t)p0 << cl; cl += l0; cw |= (uint64_t)p1 << cl; cl += l1;
                cw |= (uint64_t)s0 << cl; cl += sl0; cw |= (uint64_t)s1 << cl; cl += sl1;
            }
        }
        const uint32_t v_incl = wave_incl_scan(cl, lane);
        or_bits64(vlc_raw, vlc_bits + v_incl - cl, cw, cl);
        vlc_bits += __shfl(v_incl, 63);
        // ---- MEL events (wave-uniform) ----------------------------------------------------------
        const bool ev = active && c_q == 0;
        const bool xev = (qy == 0) && (qx & 1) && u > 0 && u_p > 0;
        const uint64_t H = __ballot(ev), V = __ballot(ev && rho != 0);
        const uint64_t XH = __ballot(xev), XV = __ballot(xev && min(u, u_p) > 2);
        uint64_t Hm = H, Vm = V;
        if (it == 0) {
            for (int pr = 0; pr < 16; ++pr) {
                const int l0 = 2 * pr, l1 = l0 + 1;
                if ((H >> l0) & 1) mel_event(mel, mel_buf, (int)((V >> l0) & 1), lane == 0);
                if ((H >> l1) & 1) mel_event(mel, mel_buf, (int)((V >> l1) & 1), lane == 0);
                if ((XH >> l1) & 1) mel_event(mel, mel_buf, (int)((XV >> l1) & 1), lane == 0);
            }
            Hm &= 0xFFFFFFFF00000000ull;
        }
        while (Hm) {
            const int i = __ffsll((long long)Hm) - 1;
            mel_event(mel, mel_buf, (int)((Vm >> i) & 1), lane == 0);
            Hm &= Hm - 1;
        }
        Bprev = Bcur;
    }
    __syncthreads();

    // ================= phase B (v1: serial in lane 0) ============================================
    uint8_t* out = a.slots + (size_t)gid * a.slot_bytes;
    if (lane == 0) {
        // MagSgn: forward, 0xFF -> next byte carries 7 bits (:415-454)
        uint32_t pos = 0, nb = 0, limit = 8;
        while (ms_bits - pos >= limit) {
            const uint32_t b = get_bits(ms_raw, pos, limit);
            out[nb++] = (uint8_t)b;
            pos += limit;
            limit = (b == 0xFF) ? 7 : 8;
        }
        const uint32_t rem = ms_bits - pos;
        if (rem > 0) {
            const uint32_t b = get_bits(ms_raw, pos, rem) | ((((1u << (limit - rem)) - 1u) << rem) & 0xFF);
            if (b != 0xFF) out[nb++] = (uint8_t)b;
        } else if (limit == 7) {
            nb--;
        }
        const uint32_t ms_len = nb;
        // VLC: bytes grow downwards from the end of the slot (:296-351)
        uint8_t* vend = out + a.slot_bytes - 1;
        vend[0] = 0xFF;
        uint32_t vpos = 1, vp = 0;
        int prev_gt = 1;
        uint32_t vacc = 0, vused = 0;
        for (;;) {
            const uint32_t avail = vlc_bits - vp;
            if (prev_gt && avail >= 7 && get_bits(vlc_raw, vp, 7) == 0x7F) {
                *(vend - vpos) = 0x7F; vpos++; vp += 7; prev_gt = 0;
                continue;
            }
            if (avail < 8) { vused = avail; vacc = avail ? get_bits(vlc_raw, vp, avail) : 0; break; }
            const uint32_t b = get_bits(vlc_raw, vp, 8);
            *(vend - vpos) = (uint8_t)b; vpos++; vp += 8; prev_gt = b > 0x8F;
        }
        // termination of MEL and VLC (:357-385)
        if (mel.run > 0) mel_put_bit(mel, mel_buf, 1, true);
        {
            const int mel_acc = mel.acc << mel.left;
            const int mel_mask = (0xFF << mel.left) & 0xFF;
            const int vlc_mask = 0xFF >> (8 - vused);
            if ((mel_mask | vlc_mask) != 0) {
                const int fuse = mel_acc | (int)vacc;
                if ((((fuse ^ mel_acc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && vpos > 1) {
                    mel_buf[mel.pos++] = (uint8_t)fuse;
                } else {
                    mel_buf[mel.pos++] = (uint8_t)mel_acc;
                    *(vend - vpos) = (uint8_t)vacc; vpos++;
                }
            }
        }
        // concatenate MagSgn | MEL | VLC and patch Scup (:924-935)
        for (uint32_t i = 0; i < mel.pos; ++i) out[ms_len + i] = mel_buf[i];
        const uint32_t total = ms_len + mel.pos + vpos;
        uint8_t* vdst = out + ms_len + mel.pos;
        const uint8_t* vsrc = vend - vpos + 1;
        for (uint32_t i = 0; i < vpos; ++i) vdst[i] = vsrc[i];       // vdst < vsrc: ascending copy is safe
        const uint32_t scup = mel.pos + vpos;
        out[total - 1] = (uint8_t)(scup >> 4);
        out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
        a.lengths[gid] = total;
    }
}

// ---- K4: exclusive scan of lengths (16-byte aligned slots in the arena) + gather ---------------
__global__ __launch_bounds__(1024) void scan_offsets_kernel(CompactArgs a)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (a.nblocks + 1023) / 1024;
    const uint32_t b0 = t * per, b1 = min(a.nblocks, b0 + per);
    uint64_t sum = 0;
    for (uint32_t i = b0; i < b1; ++i) sum += (a.lengths[i] + 15u) & ~15u;
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t o = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += o;
        __syncthreads();
    }
    uint64_t off = part[t] - sum;
    for (uint32_t i = b0; i < b1; ++i) { a.offsets[i] = off; off += (a.lengths[i] + 15u) & ~15u; }
    if (t == 1023) {
        a.offsets[a.nblocks] = part[1023];
        if (part[1023] > a.arena_bytes) *a.overflow_flag = 1;
    }
}
__global__ __launch_bounds__(64) void gather_kernel(CompactArgs a)
{
    const uint32_t b = blockIdx.x;
    const uint64_t off = a.offsets[b];
    const uint32_t len = a.lengths[b];
    if (off + len > a.arena_bytes) return;
    const uint4* s = reinterpret_cast<const uint4*>(a.slots + (size_t)b * a.slot_bytes);
    uint4* d = reinterpret_cast<uint4*>(a.arena + off);
    for (uint32_t i = threadIdx.x; i < (len + 15) / 16; i += 64) d[i] = s[i];
}

} // namespace

static bool g_tables_ready[16] = {false};

hipError_t launch_ht_encode(const HtArgs& a, hipStream_t s)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 16 && !g_tables_ready[dev]) {
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_enc), HT_VLC_ENC0, sizeof(HT_VLC_ENC0), 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_enc), HT_VLC_ENC1, sizeof(HT_VLC_ENC1), sizeof(HT_VLC_ENC0), hipMemcpyHostToDevice);
        if (e != hipSuccess) return e;
        g_tables_ready[dev] = true;
    }
    // raw MagSgn stream: at most (kmax+1) bits per sample; VLC: <= 24 bits per quad + slack
    const uint32_t max_bits = a.max_block_samples * (a.max_kmax + 2u);
    const uint32_t ms_words = max_bits / 32 + 8;
    const uint32_t vlc_words = 1024;
    const size_t shmem = (size_t)(ms_words + vlc_words + 64) * 4;
    const uint32_t nblocks = a.blocks_per_tile * a.ntiles;
    if (a.irreversible)
        hipLaunchKernelGGL(ht_encode_kernel<true>, dim3(nblocks), dim3(64), shmem, s, a, ms_words, vlc_words);
    else
        hipLaunchKernelGGL(ht_encode_kernel<false>, dim3(nblocks), dim3(64), shmem, s, a, ms_words, vlc_words);
    return hipGetLastError();
}

hipError_t launch_compact(const CompactArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(gather_kernel, dim3(a.nblocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

} // namespace grk_amd
