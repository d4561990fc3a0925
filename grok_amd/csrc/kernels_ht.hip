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
        if (writer && m.pos < 250) buf[m.pos] = (uint8_t)m.acc;   // valid streams stay < 128 bytes (:475)
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

// Event walker shared by the MagSgn and VLC streams (oracle/ht_wave_model.c: walk()).
// 64 lanes test 64 raw words per round for positions of the current byte phase where a stuffing
// event can happen; every event marks one OUTPUT byte index as "7 bits wide".
template <bool VLC>
__device__ __forceinline__ uint32_t walk_events(const uint32_t* raw, uint32_t nwords, uint32_t nbits,
                                                uint32_t* marks, uint32_t& last_p, int lane)
{
    uint32_t K = 0, s = 0;
    constexpr uint32_t need = VLC ? 7 : 8;
    while (s + need <= nbits) {
        const uint32_t B = s >> 5, phase = s & 7;
        const uint32_t i = B + lane;
        const uint32_t w0 = i < nwords ? raw[i] : 0u;
        const uint32_t w1 = i + 1 < nwords ? raw[i + 1] : 0u;
        const uint64_t hi = w0 | ((uint64_t)w1 << 32);
        uint64_t c = hi & (hi >> 1);
        c &= c >> 2;
        uint32_t cand;
        if constexpr (!VLC) {
            c &= c >> 4;
            cand = (uint32_t)c;
        } else {
            c &= c >> 3;
            const uint32_t wm = i == 0 ? 0xFFFFFFFFu : (i - 1 < nwords ? raw[i - 1] : 0u);
            const uint64_t lo = wm | ((uint64_t)w0 << 32);
            const uint64_t pv = (lo >> 31) & ((lo >> 30) | (lo >> 29) | (lo >> 28));
            cand = (uint32_t)c & (uint32_t)pv;
        }
        uint32_t m = 0x01010101u << phase;
        if (lane == 0) m &= 0xFFFFFFFFu << (s & 31);
        const uint32_t hit = cand & m;
        const uint64_t ballot = __ballot(hit != 0);
        if (!ballot) { s = 32 * (B + 64) + phase; continue; }
        const int L = __ffsll((long long)ballot) - 1;
        const uint32_t hl = __shfl(hit, L);
        const uint32_t p = 32 * (B + (uint32_t)L) + (uint32_t)(__ffs((int)hl) - 1);
        const uint32_t j = (p + K) >> 3;
        const uint32_t mj = VLC ? j : j + 1;
        if (lane == 0) marks[mj >> 5] |= 1u << (mj & 31);
        ++K; last_p = p;
        s = p + 15;
    }
    return K;
}

template <bool IRREV>
__global__ __launch_bounds__(64) void ht_encode_kernel(HtArgs a, uint32_t ms_words, uint32_t vlc_words, uint32_t mark_words, uint32_t vmark_words)
{
    // LDS: raw MagSgn bits | raw VLC bits | 64 zero words | 7-bit-byte bitmaps | their prefix counts | MEL bytes
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* ms_raw  = smem;
    uint32_t* vlc_raw = smem + ms_words;
    uint32_t* marks   = vlc_raw + vlc_words;
    uint32_t* vmarks  = marks + mark_words;
    uint16_t* pref    = reinterpret_cast<uint16_t*>(vmarks + vmark_words);
    uint16_t* vpref   = pref + mark_words;
    uint8_t*  mel_buf = reinterpret_cast<uint8_t*>(vpref + vmark_words);     // 256 bytes

    const int lane = threadIdx.x;
    const uint32_t gid = blockIdx.x;
    const uint32_t tile = gid / a.blocks_per_tile;
    const HtBlockDesc bd = a.blocks[gid % a.blocks_per_tile];
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    const int32_t* src = a.mallat + ((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px;
    const uint32_t p = 30u - bd.kmax;
    const bool vec2 = ((bd.px | a.stride) & 1u) == 0;      // 8-byte row-pair loads are aligned

    for (uint32_t i = lane; i < ms_words + vlc_words; i += 64) smem[i] = 0;
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
        int32_t rawv[4] = {0, 0, 0, 0};          // [0]=(x0,y0) [1]=(x0,y0+1) [2]=(x0+1,y0) [3]=(x0+1,y0+1)
        {
            const uint32_t x0 = 2 * qx, y0 = 2 * qy;
            if (x0 < w && y0 < h) {
                const int32_t* row0 = src + (size_t)y0 * a.stride + x0;
                const bool pair = x0 + 1 < w;
                if (vec2 && pair) { const int2 q = *reinterpret_cast<const int2*>(row0); rawv[0] = q.x; rawv[2] = q.y; }
                else { rawv[0] = row0[0]; if (pair) rawv[2] = row0[1]; }
                if (y0 + 1 < h) {
                    const int32_t* row1 = row0 + a.stride;
                    if (vec2 && pair) { const int2 q = *reinterpret_cast<const int2*>(row1); rawv[1] = q.x; rawv[3] = q.y; }
                    else { rawv[1] = row1[0]; if (pair) rawv[3] = row1[1]; }
                }
            }
        }
        uint32_t tw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int32_t raw = rawv[i];
            uint32_t t;
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

    // ================= phase B: stuffing, termination, emission (see oracle/ht_wave_model.c) =====
    const uint32_t msw = ms_words, vw = vlc_words;          // raw buffers are followed by zeroed words
    for (uint32_t i = lane; i < mark_words + vmark_words; i += 64) marks[i] = 0;
    __syncthreads();

    // ---- B1: MagSgn events
    uint32_t last_p = 0;
    const uint32_t K = walk_events<false>(ms_raw, msw, ms_bits, marks, last_p, lane);
    uint32_t pos, limit, nfull;
    if (K && last_p + 15 > ms_bits) { pos = last_p + 8; limit = 7; nfull = (pos + K - 1) >> 3; }
    else { const uint32_t s0 = K ? last_p + 15 : 0; pos = s0 + 8 * ((ms_bits - s0) >> 3); limit = 8; nfull = (pos + K) >> 3; }
    const uint32_t rem = ms_bits - pos;
    uint32_t ms_len, final_byte = 0, has_final = 0;
    if (rem > 0) {
        final_byte = get_bits(ms_raw, pos, rem) | ((((1u << (limit - rem)) - 1u) << rem) & 0xFF);
        has_final = final_byte != 0xFF;
        ms_len = nfull + has_final;
    } else {
        ms_len = (limit == 7) ? nfull - 1 : nfull;
    }
    const uint32_t ms_emit = min(nfull, ms_len);

    // ---- B2: VLC events + tail
    uint32_t vlast = 0;
    const uint32_t Kv = walk_events<true>(vlc_raw, vw, vlc_bits, vmarks, vlast, lane);
    const uint32_t vs0 = Kv ? vlast + 7 : 0;
    const uint32_t vposr = vs0 + 8 * ((vlc_bits - vs0) >> 3);
    const uint32_t vused = vlc_bits - vposr;
    const uint32_t vacc = vused ? get_bits(vlc_raw, vposr, vused) : 0;
    const uint32_t nv = (vposr + Kv) >> 3;

    // ---- B3: MEL / VLC termination (terminate_mel_vlc :357-385), wave-uniform
    if (mel.run > 0) mel_put_bit(mel, mel_buf, 1, lane == 0);
    uint32_t vextra = 0;
    {
        const int macc = mel.acc << mel.left;
        const int mel_mask = (0xFF << mel.left) & 0xFF;
        const int vlc_mask = 0xFF >> (8 - (int)vused);
        if ((mel_mask | vlc_mask) != 0) {
            const int fuse = macc | (int)vacc;
            if ((((fuse ^ macc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && nv >= 1) {
                if (lane == 0 && mel.pos < 250) mel_buf[mel.pos] = (uint8_t)fuse;
            } else {
                if (lane == 0 && mel.pos < 250) mel_buf[mel.pos] = (uint8_t)macc;
                vextra = 1;
            }
            mel.pos++;
        }
    }
    const uint32_t mel_len = mel.pos;
    const uint32_t vcount = nv + vextra;
    const uint32_t total = ms_len + mel_len + vcount + 1;
    const uint32_t scup = mel_len + vcount + 1;

    // ---- B4: prefix popcounts of the mark bitmaps (one pass covers both bitmaps back to back)
    {
        uint32_t base = 0;
        for (uint32_t w0 = 0; w0 < mark_words; w0 += 64) {
            const uint32_t i = w0 + lane;
            const uint32_t c = i < mark_words ? __popc(marks[i]) : 0;
            const uint32_t incl = wave_incl_scan(c, lane);
            if (i < mark_words) pref[i] = (uint16_t)(base + incl - c);
            base += __shfl(incl, 63);
        }
        base = 0;
        for (uint32_t w0 = 0; w0 < vmark_words; w0 += 64) {
            const uint32_t i = w0 + lane;
            const uint32_t c = i < vmark_words ? __popc(vmarks[i]) : 0;
            const uint32_t incl = wave_incl_scan(c, lane);
            if (i < vmark_words) vpref[i] = (uint16_t)(base + incl - c);
            base += __shfl(incl, 63);
        }
    }

    // ---- allocate the block's bytes in the arena (16-byte aligned, order of arrival)
    unsigned long long base_off = 0;
    if (lane == 0) base_off = atomicAdd(a.cursor, (unsigned long long)((total + 15u) & ~15u));
    base_off = ((unsigned long long)__shfl((uint32_t)(base_off >> 32), 0) << 32) | __shfl((uint32_t)base_off, 0);
    if (lane == 0) { a.lengths[gid] = total; a.offsets[gid] = base_off; }
    if (base_off + total > a.arena_bytes) {
        if (lane == 0) *a.overflow_flag = 1;
        return;
    }
    uint8_t* out = a.arena + base_off;
    __syncthreads();

    // ---- B5: emission. MagSgn: one dword (4 bytes) per lane and iteration, coalesced stores
    for (uint32_t j = 4 * lane; j < ms_emit; j += 256) {
        const uint32_t wd = j >> 5, bit = j & 31;
        const uint32_t mw = marks[wd];
        const uint32_t k = pref[wd] + __popc(mw & ((1u << bit) - 1u));
        const uint32_t flags = (mw >> bit) & 0xF;
        uint32_t start = 8 * j - k;
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const uint32_t n = ((flags >> b) & 1) ? 7u : 8u;
            word |= get_bits(ms_raw, start, n) << (8 * b);
            start += n;
        }
        if (j + 4 <= ms_emit) *reinterpret_cast<uint32_t*>(out + j) = word;
        else for (uint32_t b = 0; j + b < ms_emit; ++b) out[j + b] = (uint8_t)(word >> (8 * b));
    }
    if (has_final && lane == 0) out[ms_len - 1] = (uint8_t)final_byte;
    for (uint32_t i = lane; i < mel_len; i += 64) out[ms_len + i] = mel_buf[i < 250 ? i : 249];
    // VLC bytes are stored in reverse order of generation; the first one carries Scup's low nibble
    for (uint32_t j = lane; j < nv; j += 64) {
        const uint32_t wd = j >> 5, bit = j & 31;
        const uint32_t mw = vmarks[wd];
        const uint32_t k = vpref[wd] + __popc(mw & ((1u << bit) - 1u));
        uint32_t byte = get_bits(vlc_raw, 8 * j - k, ((mw >> bit) & 1) ? 7u : 8u);
        if (j == 0) byte = (byte & 0xF0) | (scup & 0xF);
        out[total - 2 - j] = (uint8_t)byte;
    }
    if (lane == 0) {
        if (vextra) out[total - 2 - nv] = (uint8_t)(nv == 0 ? ((vacc & 0xF0) | (scup & 0xF)) : vacc);
        out[total - 1] = (uint8_t)(scup >> 4);
    }
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
    // raw MagSgn stream: at most (kmax+1) bits per sample; VLC: < 32 bits per quad pair.
    const uint32_t max_bits = a.max_block_samples * (a.max_kmax + 2u);
    const uint32_t ms_words = max_bits / 32 + 8;
    const uint32_t vlc_bits_max = 16u * ((a.max_block_samples + 3) / 4 + 64) + 64;
    const uint32_t vlc_words = vlc_bits_max / 32 + 8;
    const uint32_t mark_words = (max_bits / 7 + 64) / 32 + 2;
    const uint32_t vmark_words = (vlc_bits_max / 7 + 64) / 32 + 2;
    const size_t shmem = (size_t)(ms_words + vlc_words + mark_words + vmark_words) * 4 +
                         (size_t)(mark_words + vmark_words + 2) * 2 + 256;
    const uint32_t nblocks = a.blocks_per_tile * a.ntiles;
    if (a.irreversible)
        hipLaunchKernelGGL(ht_encode_kernel<true>, dim3(nblocks), dim3(64), shmem, s, a, ms_words, vlc_words, mark_words, vmark_words);
    else
        hipLaunchKernelGGL(ht_encode_kernel<false>, dim3(nblocks), dim3(64), shmem, s, a, ms_words, vlc_words, mark_words, vmark_words);
    return hipGetLastError();
}

} // namespace grk_amd
