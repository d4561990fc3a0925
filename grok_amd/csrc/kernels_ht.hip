// grok_amd/csrc/kernels_ht.hip -- K3: HTJ2K cleanup-pass encoder, one wavefront per code-block.
//
// Replaces T1HT::preCompress + ojph_encode_codeblock (t1/t1_ht/T1HT.cpp:58-128,
// t1/t1_ht/coding/ojph_block_encoder.cpp:463-938), which the reference runs one block per CPU
// thread and strictly serially inside the block.
//
// The kernel is bound by VALU issue (a wave64 instruction occupies a SIMD for 4 cycles), so the
// whole of phase A is written branch-free and counted in instructions:
//   * lane <-> quad, two quad rows of 32 quads per iteration; samples of the next iteration are
//     prefetched while the current one is analysed;
//   * exponents are kept in "leading-zero" form (v_ffbh), the 4-neighbour maximum of the row
//     above is one v_pk_min_u16, the VLC table is indexed directly by (eps, rho, neighbour
//     insignificance flags) -- the table is permuted on the host so no context value is built;
//   * one packed DPP prefix sum per iteration yields the bit offsets of both the MagSgn and the
//     VLC stream; a quad's four MagSgn values are concatenated in registers and OR-ed into the
//     raw (un-stuffed) stream in LDS with three ds_or; the VLC/UVLC bits of a quad pair are
//     placed by each lane for its own quad (<= 30 bits per pair);
//   * MEL events are gathered with v_cmp (ballot) and run through the 13-state MEL coder on the
//     scalar unit.
//   phase B: byte-stuffing + termination + concatenation MagSgn | MEL | VLC(reversed) + Scup
//     (speculative windows; the forms of rounds 1, 3 and 5 are modelled on the CPU in oracle/ht_wave_model.c).
//   r05: packed 8-bit content is coded TWO quads per lane (phase_a2), pipelined encodes launch a ROOM instance that leaves the
//     next frame's DWT registers, the VLC stream is emitted once (forwards, staged in LDS): profiles/r05_k3_pairs.txt.
//
// Bit-exactness contract: the byte string per block equals ojph_encode_codeblock's for every
// input with |coefficient| < 2^(Kmax+1) (guaranteed by the BIBO-derived exponents for in-range
// pixels); larger magnitudes raise GRK_AMD_ERR_UNSUPPORTED instead of producing other bytes.
#include <atomic>
#include <mutex>
#include "kernels.h"
#include "ht_vlc_tables.h"
// the coded bytes are written once and read by nobody on the device: non-temporal stores
#define GRK_K3_STORE(p, v) __builtin_nontemporal_store((uint32_t)(v), (p))
#include <type_traits>

namespace grk_amd {

namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

// MEL exponents E[k], k = 0..12 = {0,0,0,1,1,1,2,2,2,3,3,4,5} (ojph_block_encoder.cpp:226), one nibble each
constexpr uint64_t kMelE = 0x5433222111000ull;

// device tables (filled by launch_ht_encode on first use per device)
//   g_vlc_enc[0..2047]    first quad row : index (c_q<<8)|(rho<<4)|eps
//   g_vlc_enc[2048..4095] other rows     : index (ctx<<8)|(rho<<4)|eps with ctx = n_w | rl<<1 | n_e<<2,
//                                          n_w/n_e = "both upper neighbours insignificant" flags
//   entry = cwd << 25 | len << 4 | e_k spread as the kernel's byte-wise arithmetic wants it: bit 0 of byte i = sample i
__device__ uint32_t g_vlc_enc[4096];
//   g_uvlc[u] : x = pre<<8 | suf<<16, y = pre_len<<8 | suf_len<<16 (ojph_block_encoder.cpp:189-210);
//   entries 33,34: the 1-bit code (u-1) of the second quad in the first-row "u0>2, u1 in 1..2" mode
__device__ uint2 g_uvlc[64];

// The MEL coder runs on the scalar unit throughout -- its state never touches a vector register: the finished bytes collect in
// ONE vector register, dword j in lane j (v_writelane from scalar registers), and reach LDS once, at the end (mel_flush).  With
// lane 0 storing every finished byte to LDS (address and data of a ds_write are vector operands) the compiler kept the whole
// state machine in vector registers: 3 363 vector instructions per block against 3 088 now, 1 556 scalar against 1 894.
// (r03 measured this form at K3 = 0.34 ms, 27 % of a wave's time in s_waitcnt, and found no difference; at 0.27 ms, the waits
//  gone and the kernel bound by its vector issue, it is 3-4 % of K3 and 1.5-2 % of the pipelined frame:
//  profiles/r04_k3_scalar_mel.txt.)
struct MelState {
    int run, k, acc, left;
    uint32_t pos;
    uint32_t word;           // the bytes of dword pos / 4 finished so far (scalar)
    uint32_t vec;            // lane j: dword j of the MEL bytes
};
// v_writelane_b32 through its LLVM intrinsic (this clang has no builtin for it; the compiler places the lane select in M0 itself)
extern "C" __device__ int grk_amd_writelane(int value, int lane_select, int old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ void mel_writelane(uint32_t& vec, uint32_t word, uint32_t lane_index)
{
    vec = (uint32_t)grk_amd_writelane(__builtin_amdgcn_readfirstlane((int)word), __builtin_amdgcn_readfirstlane((int)lane_index), (int)vec);
}
// Between the quad rows the coder's state travels packed -- run (6 bits) | k << 6 | acc << 10 | left << 18 | pos << 22, the unfinished
// dword, the bytes' vector register, and a queue of up to 60 bits (MSB first) that have not been through the byte packer yet -- and a row
// touches what it needs: no events, nothing; events, run / k and the queue (mel_first_row, mel_row); the byte packer only when 40 bits are
// queued.  Carried as six scalars through the unrolled rows the state cost every row six copies and three v_readfirstlane (the compiler's
// SGPR-copy pass moves a state word that meets a vector value in a phi to the vector unit), ~28 instructions per row without a single
// event; unpacked, drained and packed again by every row WITH events (the first form of this) ~50 instructions per such row, on
// quantised content 12 % of K3's instructions.
struct MelPacked { uint32_t st, word, vec; uint64_t q; uint32_t qn; };
__device__ __forceinline__ void mel_drain(MelState& m, uint64_t q, uint32_t qn);
__device__ __forceinline__ MelState mel_unpack(const MelPacked& p)         // the whole state, the queue drained into it
{
    const uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.st);
    MelState m;
    m.run = (int)(st & 63u); m.k = (int)((st >> 6) & 15u); m.acc = (int)((st >> 10) & 255u); m.left = (int)((st >> 18) & 15u);
    m.pos = st >> 22; m.word = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.word); m.vec = p.vec;
    const uint64_t q = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.q >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.q);
    mel_drain(m, q, (uint32_t)__builtin_amdgcn_readfirstlane((int)p.qn));
    return m;
}
__device__ __forceinline__ void mel_pack(MelPacked& p, const MelState& m)
{
    p.st = (uint32_t)m.run | ((uint32_t)m.k << 6) | (((uint32_t)m.acc & 255u) << 10) | ((uint32_t)m.left << 18) | (m.pos << 22);
    p.word = m.word; p.vec = m.vec; p.q = 0; p.qn = 0;
}
__device__ __forceinline__ void mel_put_byte(MelState& m, uint32_t byte)
{
    if (m.pos < 252) {
        m.word |= byte << (8u * (m.pos & 3u));
        if ((m.pos & 3u) == 3u) { mel_writelane(m.vec, m.word, m.pos >> 2); m.word = 0; }
    }
    m.pos++;
}
__device__ __forceinline__ void mel_put_bit(MelState& m, uint8_t*, int v, bool)
{
    m.acc = (m.acc << 1) | v;
    if (--m.left == 0) {
        mel_put_byte(m, (uint32_t)m.acc & 0xFFu);
        m.left = (m.acc == 0xFF) ? 7 : 8;
        m.acc = 0;
    }
}
// n <= 6 bits (MSB first) at once: at most one byte is completed on the way (a byte has 8 bits, 7 behind a 0xFF)
__device__ __forceinline__ void mel_put_bits(MelState& m, uint32_t v, uint32_t n)
{
    if ((int)n < m.left) { m.acc = (m.acc << n) | (int)v; m.left -= (int)n; return; }
    const uint32_t rem = n - (uint32_t)m.left;
    const uint32_t byte = (((uint32_t)m.acc << m.left) | (v >> rem)) & 0xFFu;
    mel_put_byte(m, byte);
    m.left = (byte == 0xFFu ? 7 : 8) - (int)rem;
    m.acc = (int)(v & ((1u << rem) - 1u));
}
__device__ __forceinline__ void mel_flush(MelState& m, uint8_t* buf, int lane)
{
    if (m.pos < 252 && (m.pos & 3u)) mel_writelane(m.vec, m.word, m.pos >> 2);
    reinterpret_cast<uint32_t*>(buf)[lane] = m.vec;
}

__device__ __forceinline__ void mel_event(MelState& m, uint8_t* buf, int one, bool writer)
{
    const int e = (int)((kMelE >> (4 * m.k)) & 0xF);
    if (!one) {
        if (++m.run >= (1 << e)) {
            mel_put_bit(m, buf, 1, writer);
            m.run = 0;
            if (m.k < 12) m.k++;
        }
    } else {
        // '0', then the e bits of the run so far (run < 2^e): e + 1 <= 6 bits in one piece
        mel_put_bits(m, (uint32_t)m.run, (uint32_t)e + 1u);
        m.run = 0;
        if (m.k > 0) m.k--;
    }
}

// qn queued bits (q, MSB first) to the byte packer, a byte at a time: a byte takes m.left more bits (8, 7 behind a 0xFF, less what it
// already holds)
__device__ __forceinline__ void mel_drain(MelState& m, uint64_t q, uint32_t qn)
{
    while ((int)qn >= m.left) {
        qn -= (uint32_t)m.left;
        const uint32_t byte = (((uint32_t)m.acc << m.left) | ((uint32_t)(q >> qn) & ((1u << m.left) - 1u))) & 0xFFu;
        mel_put_byte(m, byte);
        m.left = byte == 0xFFu ? 7 : 8;
        m.acc = 0;
    }
    m.acc = (int)(((uint32_t)m.acc << qn) | ((uint32_t)q & ((1u << qn) - 1u)));
    m.left -= (int)qn;
}
// ... of a packed state: the byte packer's fields of st (acc, left, pos), the unfinished dword, the vector register; run and k stay
__device__ __forceinline__ void mel_drain_packed(uint32_t& st, MelPacked& p, uint64_t& q, uint32_t& qn)
{
    MelState m;
    m.run = 0; m.k = 0; m.acc = (int)((st >> 10) & 255u); m.left = (int)((st >> 18) & 15u);
    m.pos = st >> 22; m.word = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.word); m.vec = p.vec;
    mel_drain(m, q, qn);
    st = (st & 0x3FFu) | (((uint32_t)m.acc & 255u) << 10) | ((uint32_t)m.left << 18) | (m.pos << 22);
    p.word = m.word; p.vec = m.vec;
    q = 0; qn = 0;
}
// nz zero events (run / k4 = 4 k / queue in registers): a '1' per completed run of 2^E[k] -- one trip per completed run, not per
// event (at most 14 from 64 zeros)
__device__ __forceinline__ void mel_zeros(uint32_t& run, uint32_t& k4, uint64_t& q, uint32_t& qn, uint32_t nz)
{
    while (true) {
        const uint32_t need = (1u << ((uint32_t)(kMelE >> k4) & 0xFu)) - run;
        if (nz < need) { run += nz; break; }
        q = (q << 1) | 1u; qn += 1u;
        run = 0; k4 = min(k4 + 4u, 48u); nz -= need;
    }
}
// The first quad row's events (EH: lanes with an event, in event order; EV: its value), from the coder's initial state.  On dense
// content ~17 events of mixed value -- the first quad's and the u-event of the 16 pairs; through mel_zero_run + mel_event (a
// run-skipping loop made for long runs of zeros) they were ~50 scalar instructions and three taken branches each, 15 % of a block's
// instructions (profiles/r05_k3_pairs.txt 4, 14).  Here an event is 16 instructions without a branch -- one: '0' and the E[k] bits of
// the run so far; zero: '1' when it completes the run, else nothing; written out on the scalar unit: from the C form of these selects the
// compiler made 35-50 instructions and up to three branches -- and its bits go to the state's queue: at most 32 events (a quad behind
// a significant one has no event of its own, a pair with a u-event has two significant quads), at most 47 bits from the initial state.
__device__ __forceinline__ void mel_first_row(MelPacked& p, uint64_t EH, uint64_t EV)
{
    uint32_t run = 0, k4 = 0, qn = 0;
    uint64_t q = 0;
    if (!(EH & EV)) mel_zeros(run, k4, q, qn, (uint32_t)__builtin_popcountll(EH));
    else while (EH) {
        uint32_t e = (uint32_t)(kMelE >> k4) & 0xFu;
        uint32_t n, v, at, t0, t1, t2;
        asm volatile("s_ff1_i32_b64 %[at], %[EH]\n\t"              // the next event's lane
                     "s_bitset0_b64 %[EH], %[at]\n\t"
                     "s_add_i32 %[t1], %[run], 1\n\t"
                     "s_lshr_b32 %[t0], %[t1], %[e]\n\t"           // full: run + 1 <= 2^e, 1 when a zero completes the run
                     "s_add_i32 %[t2], %[t0], -1\n\t"
                     "s_and_b32 %[t1], %[t1], %[t2]\n\t"           // the run behind a zero: 0 when complete, else run + 1
                     "s_lshl2_add_u32 %[t2], %[t0], %[k4]\n\t"
                     "s_min_i32 %[t2], %[t2], 48\n\t"             // k behind a zero: + 1 when the run is complete (<= 12)
                     "s_add_i32 %[k4], %[k4], -4\n\t"
                     "s_max_i32 %[k4], %[k4], 0\n\t"              // k behind a one: - 1 (>= 0)
                     "s_add_i32 %[e], %[e], 1\n\t"
                     "s_bitcmp1_b64 %[EV], %[at]\n\t"             // SCC = the event's value
                     "s_cselect_b32 %[n], %[e], %[t0]\n\t"        // one: '0' + the e bits of the run so far; zero: '1' when complete
                     "s_cselect_b32 %[v], %[run], %[t0]\n\t"
                     "s_cselect_b32 %[run], 0, %[t1]\n\t"
                     "s_cselect_b32 %[k4], %[k4], %[t2]"
                     : [EH] "+s"(EH), [run] "+s"(run), [k4] "+s"(k4), [e] "+s"(e), [n] "=&s"(n), [v] "=&s"(v), [at] "=&s"(at),
                       [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2)
                     : [EV] "s"(EV)
                     : "scc");
        q = (q << n) | v; qn += n;
    }
    p.st = (p.st & ~0x3FFu) | run | (k4 << 4);                     // (k << 6)
    p.q = q; p.qn = qn;
}

// A later quad row's events, where zeros are the rule (H: quads with an event, V: its value): the zeros up to the next one are one
// addition unless they complete a run, then that one's bits go to the queue.  (Quantised 16-bit content, BASELINE configs[2]: through
// mel_zero_run + mel_event, a byte packer behind every bit, these rows were 14.5 % of K3's time -- a what-if build without them 0.362
// against 0.423 ms.)
__device__ __forceinline__ void mel_row(MelPacked& p, uint64_t H, uint64_t V)
{
    uint32_t st = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.st);
    uint32_t run = st & 63u, k4 = (st >> 4) & 0x3Cu;
    uint64_t q = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.q >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.q);
    uint32_t qn = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.qn);
    if (__builtin_expect(qn > 40u, 0)) mel_drain_packed(st, p, q, qn);
    uint64_t ones = H & V;
    while (ones) {
        const uint32_t first = (uint32_t)__builtin_ctzll(ones);
        const uint64_t from = ~0ull << first;                       // the one's lane and above
        const uint32_t nz = (uint32_t)__builtin_popcountll(H & ~from);
        H &= from << 1; ones &= from << 1;
        uint32_t e = (uint32_t)(kMelE >> k4) & 0xFu;
        if (__builtin_expect(run + nz >= (1u << e), 0)) { mel_zeros(run, k4, q, qn, nz); e = (uint32_t)(kMelE >> k4) & 0xFu; }
        else run += nz;
        q = (q << (e + 1u)) | run; qn += e + 1u;                   // the one: '0' and the E[k] bits of the run so far
        run = 0; k4 = (uint32_t)max((int)k4 - 4, 0);
        if (__builtin_expect(qn > 40u, 0)) mel_drain_packed(st, p, q, qn);   // (one trip adds at most 14 + 6 bits)
    }
    if (H) mel_zeros(run, k4, q, qn, (uint32_t)__builtin_popcountll(H));     // the zeros behind the last one
    p.st = (st & ~0x3FFu) | run | (k4 << 4);
    p.q = q; p.qn = qn;
}

// n events "quad not significant" in a row: a run of 2^E[k] of them is one 1 bit, so the loop runs once per emitted bit, not once
// per event -- an all-zero 64 x 64 block is 1 024 such events and ~45 bits (r04: K3 on an all-zero 4096^2 frame 0.68 ms, seven
// times the time of real content, because every event went through the state machine by itself)
__device__ __forceinline__ void mel_zero_run(MelState& m, uint8_t* buf, uint32_t n, bool writer)
{
    while (n) {
        const uint32_t need = (1u << ((kMelE >> (4 * m.k)) & 0xF)) - (uint32_t)m.run;
        if (n < need) { m.run += (int)n; return; }
        mel_put_bit(m, buf, 1, writer);
        m.run = 0;
        if (m.k < 12) m.k++;
        n -= need;
    }
}

// acc |= a | b, HERE: a plain `acc |= a | b` in straight-line code is re-associated into one OR tree at the accumulator's only use (the
// range check behind phase A), and every magnitude word of the block -- 32 registers in the pair form -- stays live until then
__device__ __forceinline__ void or3_now(uint32_t& acc, uint32_t a, uint32_t b)
{
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xfe" : "=v"(acc) : "v"(acc), "v"(a), "v"(b));      // (a | b | c at the 2-cycle rate; v_or3_b32 takes 4)
}
__device__ __forceinline__ void lds_or(uint32_t* p, uint32_t v)
{
    (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// OR val (no bits at or above 64) into the little-endian bit array at bit `pos`
__device__ __forceinline__ void or_bits64(uint32_t* raw, uint32_t pos, uint64_t val, uint32_t nbits)
{
    const uint32_t sh = pos & 31;
    uint32_t* w = raw + (pos >> 5);
    const uint64_t lo = val << sh;
    lds_or(w, (uint32_t)lo);
    lds_or(w + 1, (uint32_t)(lo >> 32));
    // the value (nbits long) reaches into a third word only when it is longer than 64 - sh bits: for most iterations of most blocks no
    // lane's does (a quad of 8-bit content has ~13 MagSgn bits), and an add + compare + scalar branch is a quarter of a ds_or
    // (r05: tested on the length instead of on the third word itself, which then is not computed at all -- K3 -0.9 %)
    if (__ballot(sh + nbits > 64u)) lds_or(w + 2, ((uint32_t)(val >> 32) >> 1) >> (31 - sh));
}
__device__ __forceinline__ void or_bits32(uint32_t* raw, uint32_t pos, uint32_t val)
{
    const uint32_t sh = pos & 31;
    uint32_t* w = raw + (pos >> 5);
    const uint64_t lo = (uint64_t)val << sh;
    lds_or(w, (uint32_t)lo);
    lds_or(w + 1, (uint32_t)(lo >> 32));
}
// n <= 25 bits starting at bit `pos` of a little-endian bit array
__device__ __forceinline__ uint32_t get_bits(const uint32_t* raw, uint32_t pos, uint32_t n)
{
    const uint32_t w = pos >> 5, sh = pos & 31;
    uint64_t v = raw[w] | ((uint64_t)raw[w + 1] << 32);
    return (uint32_t)(v >> sh) & ((1u << n) - 1);
}

// v_ffbh_i32: leading bits equal to the sign bit (= clz for positive input), 0xFFFFFFFF for 0 and -1
__device__ __forceinline__ uint32_t ffbh_i32(uint32_t t)
{
    uint32_t r;
    asm("v_ffbh_i32 %0, %1" : "=v"(r) : "v"(t));
    return r;
}

// v_ffbh_u32: leading zeros, 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t ffbh_u32(uint32_t t)
{
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(t));
    return r;
}
// leading-zero count (v_ffbh_u32 of a 16-bit half, or v_ffbh_i32 of a word) written into byte B of `acc`, the other
// bytes kept: four of them leave a quad's exponents packed in one register without a single unpack / pack instruction
template <int B, int HALF>
__device__ __forceinline__ void ffbh_u16_to_byte(uint32_t& acc, uint32_t packed)
{
    if constexpr (B == 0) {
        if constexpr (HALF == 0) asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(acc) : "v"(packed));
        else asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(acc) : "v"(packed));
    } else if constexpr (B == 1) {
        if constexpr (HALF == 0) asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(acc) : "v"(packed));
        else asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(acc) : "v"(packed));
    } else if constexpr (B == 2) {
        if constexpr (HALF == 0) asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(acc) : "v"(packed));
        else asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(acc) : "v"(packed));
    } else {
        if constexpr (HALF == 0) asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(acc) : "v"(packed));
        else asm("v_ffbh_u32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(acc) : "v"(packed));
    }
}
template <int B>
__device__ __forceinline__ void ffbh_i32_to_byte(uint32_t& acc, uint32_t t)
{
    if constexpr (B == 0) asm("v_ffbh_i32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(acc) : "v"(t));
    else if constexpr (B == 1) asm("v_ffbh_i32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(t));
    else if constexpr (B == 2) asm("v_ffbh_i32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(t));
    else asm("v_ffbh_i32_sdwa %0, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(acc) : "v"(t));
}
__device__ __forceinline__ uint32_t min_of_bytes(uint32_t c)
{
    uint32_t m1, m2;
    asm("v_min_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "=v"(m1) : "v"(c));
    asm("v_min_u32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3" : "=v"(m2) : "v"(c));
    return min(m1, m2);
}
// any function of three bit vectors in one 2-cycle instruction; bit (a << 2 | b << 1 | c) of TT is the result for inputs a, b, c:
// 0xEA (a & b) | c   0xA8 (a | b) & c   0xE4 c ? a : b   0x80 a & b & c   0x30 a & ~b
template <int TT>
__device__ __forceinline__ uint32_t bitop3(uint32_t a, uint32_t b, uint32_t c)
{
    return (uint32_t)__builtin_amdgcn_bitop3_b32((int)a, (int)b, (int)c, TT);
}
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// (written as instructions: the compiler rewrites the C++ forms of these three into compare + select sequences)
__device__ __forceinline__ uint32_t pk_lshr15_u16(uint32_t a)          // bit 15 of each half -> bit 0 of that half
{
    uint32_t r;
    asm("v_pk_lshrrev_b16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t pk_sub_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sign_mask(uint32_t a)               // 0xFFFFFFFF when bit 31 is set, else 0
{
    uint32_t r;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t pk_mul_lo_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, true);
}
// inclusive prefix sum over the 64 lanes: 4 row_shr steps inside each row of 16, then row_bcast
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp0<0x111, 0xF>(v);      // row_shr:1
    v += dpp0<0x112, 0xF>(v);      // row_shr:2
    v += dpp0<0x114, 0xF>(v);      // row_shr:4
    v += dpp0<0x118, 0xF>(v);      // row_shr:8
    v += dpp0<0x142, 0xA>(v);      // row_bcast:15 -> rows 1,3
    v += dpp0<0x143, 0xC>(v);      // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t quad_swap(uint32_t v)          // value of lane ^ 1
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// ---- arena allocation ---------------------------------------------------------------------------
// A single device-scope cursor serialises at the memory side (measured 13.6 ns per atomic on
// MI355X = 0.67 ms for the 49 152 blocks of an 8K image, more than the coding itself), so blocks
// allocate from one of kAllocRegions region words (own cache line each) that hand out space inside
// a chunk; only a chunk refill (every ~30 blocks) touches the shared cursor.  The arena stays one
// compact extent [0, *cursor) with at most a block-sized gap at chunk ends.
// Region word: bits 63..24 = chunk start / 16, bits 23..0 = 16-byte units used in the chunk.

__global__ void ht_alloc_init_kernel(unsigned long long* flagbuf, uint32_t chunk_units)
{
    ht_alloc_reset(flagbuf, chunk_units, threadIdx.x, blockDim.x);
}

__device__ __forceinline__ unsigned long long arena_alloc(unsigned long long* flagbuf, uint32_t region, uint32_t bytes, uint32_t kChunkUnits)
{
    unsigned long long* word = flagbuf + 32 * (1 + region);
    const unsigned long long n = (bytes + 15u) >> 4;
    // (a request that does not fit a chunk -- the host sizes chunks at twice the largest block, context.hip make_ht_args: only a
    //  geometry it did not foresee gets here -- takes its bytes from the shared cursor itself and leaves the region's chunk alone;
    //  handed a fresh chunk it would run past the chunk's end into the next region's)
    if (n > kChunkUnits) return __hip_atomic_fetch_add(flagbuf + 1, n << 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (true) {
        const unsigned long long old = __hip_atomic_fetch_add(word, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long used = old & 0xFFFFFFull, start = old >> 24;
        if (used + n <= kChunkUnits) return (start + used) << 4;
        if (used <= kChunkUnits) {                              // this allocation crossed the chunk end: refill
            const unsigned long long fresh =
                __hip_atomic_fetch_add(flagbuf + 1, (unsigned long long)kChunkUnits << 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 4;
            __hip_atomic_store(word, (fresh << 24) | n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return fresh << 4;
        }
        // chunk exhausted, another wave is installing the next one
        while ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 24) == start) __builtin_amdgcn_s_sleep(8);
    }
}

// LDS words of the two raw streams, the two bitmaps, and what the raw streams may hold (bits) before the block is
// handed to the fallback launch
struct HtLds { uint32_t ms_words, vlc_words, ms_cap_bits, vlc_cap_bits, stage_bytes; };

// One code-block, by one wavefront.  li = index of the block in the launch's class list, tile = tile index.
// H16: the Mallat planes hold int16 coefficients (reversible, 8-bit pixels; kernels_dwt.hip H16)
template <bool IRREV, bool H16>
__device__ __forceinline__ void ht_encode_block(const HtArgs& a, uint32_t li, uint32_t tile, const HtLds& L, uint32_t class_id)
{
    const uint32_t ms_words = L.ms_words, vlc_words = L.vlc_words;
    // LDS (sized by the launch, HtLds: for real content rather than the worst case, so that 24 waves fit a CU): raw MagSgn bits | raw VLC bits |
    // UVLC table | MEL bytes.  (Phase B's windows read up to 65 words past the end of a raw stream: the table and the MEL bytes
    // are 192 words -- what they read there belongs to lanes whose bytes are masked out.)
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* ms_raw  = smem;
    uint32_t* vlc_raw = ms_raw + ms_words;
    uint2*    uvlc_l  = reinterpret_cast<uint2*>(vlc_raw + vlc_words);           // 64 entries (phase A only: phase B stages the VLC bytes over them)
    uint8_t*  mel_buf = reinterpret_cast<uint8_t*>(uvlc_l) + L.stage_bytes;      // 256 bytes (written once, by mel_flush, behind phase A)

    // (the lane index through an opaque move: in a kernel whose workgroups walk a list of blocks -- the fallback launch -- the compiler
    //  otherwise hoists everything derived from it out of the walk: 182 registers instead of 97)
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(threadIdx.x));
    // this launch covers the blocks sel[0..sel_count) of every tile (all blocks when sel == nullptr)
    const uint32_t lb = a.sel ? a.sel[li] : li;
    const uint32_t gid = tile * a.blocks_per_tile + lb;
    const HtBlockDesc bd = a.blocks[lb];
    const uint32_t w = bd.w, h = bd.h;
    const uint32_t QW = (w + 1) >> 1, QH = (h + 1) >> 1;
    constexpr uint32_t EB = H16 ? 2u : 4u;         // bytes per coefficient
    const char* src = reinterpret_cast<const char*>(a.mallat) +
                      (((size_t)tile * a.ncomp + bd.comp) * a.pitch + (size_t)bd.py * a.stride + bd.px) * EB;
    const bool full = w == 64 && h == 64 && ((bd.px | a.stride) & 1u) == 0;   // aligned row-pair loads, no edges
    const uint32_t kmax = bd.kmax;
    const bool narrow = kmax + 2 <= 16;           // a quad's four MagSgn values fit 64 bits

    uvlc_l[lane] = g_uvlc[lane];
    {   // clear the raw streams 16 bytes per lane and instruction (ms_words + vlc_words is a multiple of 4)
        uint4* z = reinterpret_cast<uint4*>(ms_raw);
        for (uint32_t i = lane; i < (ms_words + vlc_words) / 4; i += 64) z[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if (lane == 0) vlc_raw[0] = 0xF;             // vlc_init: four 1 bits pending (:315-318)
    __syncthreads();

    MelPacked melp{8u << 18, 0, 0, 0, 0};         // run 0, k 0, no bits, 8 to go in byte 0, nothing queued
    uint32_t ms_bits = 0, vlc_bits = 4;
    bool lds_full = false;
    uint32_t Bprev = 0xFFFFFFFFu;                // "all insignificant" row above the block
    uint32_t ovf = 0;
    const uint32_t qx = lane & 31, half = lane >> 5;
    const bool odd = lane & 1;
    const bool isq0 = qx == 0, isq31 = qx == 31;
    const uint32_t iters = (QH + 1) >> 1;
    const uint32_t stride_b = a.stride * EB;
    const char* srcb = src;
    const float inv_step = bd.inv_step;
    const uint32_t lim = (1u << kmax) - 1u;

    // Phase A exists twice: FULL for 64x64 blocks whose row pairs can be loaded as aligned 8-byte
    // words (all but the lowest resolutions), and the general edge-handling version.  The choice is
    // made once per block: a branch inside the fetch would make the loads land in merged registers
    // and be waited for on the spot, which defeats the prefetch.
    //
    // Instruction selection follows the issue rates measured on gfx950 (profiles/r02_valu_issue_rates.txt): a wave64
    // and / or / xor / not / add / sub / right shift / v_bitop3 holds the SIMD for 2 cycles, everything else (left
    // shifts, min / max, ffbh, bfe, every compare and select, SDWA, DPP, v_pk_*) for 4 -- so flags are gathered with
    // right shifts + v_bitop3 instead of compare + select, and 16-bit samples are analysed two at a time (v_pk_*).
    const uint32_t q0m = isq0 ? 0xFFFFFFFFu : 0u, q31m = isq31 ? 0xFFFFFFFFu : 0u;   // lanes without a left / right neighbour
    const uint32_t hmask = half ? 0xFFFFFFFFu : 0u;
    const uint32_t lane_off = 2u * half * stride_b + 2u * qx * EB;                  // FULL: the lane's samples inside an iteration's rows
    const char* const vtab = reinterpret_cast<const char*>(a.vlc_tab);
    auto phase_a = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    constexpr bool PK = FULL && H16;         // samples stay packed: word 0 = (x0, y0) | (x0+1, y0) << 16, word 1 = the row below

    // ---- sample fetch: r[0]=(x0,y0) [1]=(x0,y0+1) [2]=(x0+1,y0) [3]=(x0+1,y0+1) -------------------
    auto fetch = [&](uint32_t it, int32_t (&r)[4]) {
        if constexpr (FULL) {
            // 32-bit offsets from the block's (wave-uniform) origin: scalar base + vector offset addressing, one add per row
            const uint32_t o0 = lane_off + it * (4u * stride_b), o1 = o0 + stride_b;
            if constexpr (H16) {     // one word = the row's two samples
                r[0] = __builtin_nontemporal_load(reinterpret_cast<const int32_t*>(srcb + o0));
                r[1] = __builtin_nontemporal_load(reinterpret_cast<const int32_t*>(srcb + o1));
            } else {
                const i32x2 q0 = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(srcb + o0));
                const i32x2 q1 = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(srcb + o1));
                r[0] = q0.x; r[2] = q0.y; r[1] = q1.x; r[3] = q1.y;
            }
        } else {
            // addresses clamped into the block; what lies outside is zeroed when the values are used
            const uint32_t x0 = 2 * qx, y0 = 4 * it + 2 * half;
            const uint32_t xa = min(x0, w - 1), xb = min(x0 + 1, w - 1);
            const uint32_t ya = min(y0, h - 1), yb = min(y0 + 1, h - 1);
            using ET = typename std::conditional<H16, int16_t, int32_t>::type;
            r[0] = *reinterpret_cast<const ET*>(srcb + ya * stride_b + xa * EB);
            r[2] = *reinterpret_cast<const ET*>(srcb + ya * stride_b + xb * EB);
            r[1] = *reinterpret_cast<const ET*>(srcb + yb * stride_b + xa * EB);
            r[3] = *reinterpret_cast<const ET*>(srcb + yb * stride_b + xb * EB);
        }
    };

    // Samples are fetched FOUR iterations ahead, into four register sets used in turn (the loop below is unrolled by four, so no
    // set is ever copied).  A set is refilled at the END of stage 1, when the last use of its old samples is behind: issued
    // earlier, the load needs a register of its own and its result a copy into the set -- which the compiler places at the end of
    // the iteration, with a wait for the load just issued in front of it (r03: what the two-ahead form of r01 / r02 compiled to;
    // 27 % of a wave's time went into s_waitcnt).  Same box: two ahead with the copy K3 0.340 / 0.338 ms alone, pipelined step
    // 0.4665 / 0.4667; four ahead with rotating copies 0.326 / 0.323, 0.4395 / 0.4431.
    const uint32_t itn = FULL ? 16u : iters;              // (a FULL block is 64 x 64: the loop's conditions fold, and with them the
                                                          //  merges at which the compiler would wait for every load in flight)
    int32_t n0[4] = {0, 0, 0, 0}, n1[4] = {0, 0, 0, 0}, n2[4] = {0, 0, 0, 0}, n3[4] = {0, 0, 0, 0};
    fetch(0, n0);
    if (itn > 1) fetch(1, n1);
    if (itn > 2) fetch(2, n2);
    if (itn > 3) fetch(3, n3);

    // The loop is software-pipelined: stage 1 of iteration it+1 (sample analysis, neighbourhood,
    // VLC table index -> table load issued) runs before stage 2 of iteration it (everything that
    // needs the table entry), so the table latency hides behind a stage of arithmetic.
    struct Stage1 {
        uint32_t vv[4];          // MagSgn values 2 mag - 2 + sign; PK: [0] = samples 0 | 2 << 16, [1] = samples 1 | 3 << 16
        uint32_t R;              // significance, one flag per byte: bit 0 sample 0, bit 8 sample 1, bit 16 sample 2, bit 24 sample 3
        uint32_t U, u, tuple;
        uint64_t H, V;           // MEL: quads coded with context 0, and which of them are significant (ballots: scalar registers)
    };

    auto stage1r = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[4], auto refill_c) {
        // C: the four exponents as leading-zero counts of 2 mag - 1, one per byte (sample i in byte i; 0xFF: insignificant)
        uint32_t C;
        if constexpr (PK) {
            const int32_t nw0 = nbuf[0], nw1 = nbuf[1];
            const i16x2 w0 = __builtin_bit_cast(i16x2, nw0), w1 = __builtin_bit_cast(i16x2, nw1);
            const u16x2 p0 = __builtin_bit_cast(u16x2, __builtin_elementwise_max(w0, -w0));      // magnitudes
            const u16x2 p1 = __builtin_bit_cast(u16x2, __builtin_elementwise_max(w1, -w1));
            ovf |= __builtin_bit_cast(uint32_t, p0) | __builtin_bit_cast(uint32_t, p1);
            const u16x2 one2 = {1, 1};
            const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(p0 + p0, one2));   // 2 mag - 1, 0: insignificant
            const uint32_t t1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(p1 + p1, one2));
            ffbh_u16_to_byte<0, 0>(C, t0); ffbh_u16_to_byte<1, 0>(C, t1);
            ffbh_u16_to_byte<2, 1>(C, t0); ffbh_u16_to_byte<3, 1>(C, t1);
            // 2 mag - 2 + sign = (2 mag - 1) - (1 - sign); 1 - sign = bit 15 of ~w per half
            o.vv[0] = pk_sub_u16(t0, pk_lshr15_u16(~(uint32_t)nw0));
            o.vv[1] = pk_sub_u16(t1, pk_lshr15_u16(~(uint32_t)nw1));
        } else {
            int32_t r[4] = {nbuf[0], nbuf[1], nbuf[2], nbuf[3]};
            if constexpr (!FULL) {
                const uint32_t x0 = 2 * qx, y0 = 4 * it + 2 * half;
                const bool ox0 = x0 < w, ox1 = x0 + 1 < w, oy0 = y0 < h, oy1 = y0 + 1 < h;
                r[0] = (ox0 && oy0) ? r[0] : 0; r[2] = (ox1 && oy0) ? r[2] : 0;
                r[1] = (ox0 && oy1) ? r[1] : 0; r[3] = (ox1 && oy1) ? r[3] : 0;
            }
            // ---- per-sample analysis (:513-563): magnitude, MagSgn value, exponent as leading-zero count
            uint32_t t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t mag;
                if constexpr (IRREV) {
                    const float cf = __int_as_float(r[i]);
                    const float q = __fmul_rn(fabsf(cf), inv_step);
                    mag = min((uint32_t)q, lim);
                } else {
                    mag = (uint32_t)max(r[i], -r[i]);
                }
                ovf |= mag;
                t[i] = (mag << 1) - 1u;                                // val - 1  (val = 2*mag); -1: insignificant
                o.vv[i] = t[i] - 1u + ((uint32_t)r[i] >> 31);          // val - 2 + sign
            }
            ffbh_i32_to_byte<0>(C, t[0]); ffbh_i32_to_byte<1>(C, t[1]);     // clz(val-1); 0xFF when insignificant
            ffbh_i32_to_byte<2>(C, t[2]); ffbh_i32_to_byte<3>(C, t[3]);
        }
        const uint32_t N = bitop3<0x30>(0x01010101u, C >> 7, 0u);      // significance, one flag per byte (a & ~b)
        const uint32_t rho = __builtin_amdgcn_udot4(N, 0x08040201u, 0u, false);
        const uint32_t rho2 = __builtin_amdgcn_udot4(N, 0x20100804u, 0u, false);       // rho << 2
        const uint32_t cm = min_of_bytes(C);
        const uint32_t emax = 32u - min(cm, 32u);

        // ---- neighbourhood: exponents / significance of the sample row above, left quad's rho ----
        const uint32_t Bcur = __builtin_amdgcn_perm(C, C, 0x0C030C01u);         // lo16 = byte 1 (sample 1), hi16 = byte 3 (sample 3)
        // the row above lives in the other half of the wave: this iteration's upper quad row (half 0) takes the lower row of
        // the iteration before from lanes 32..63 (Bprev), the lower quad row (half 1) this iteration's upper row from lanes
        // 0..31 (Bcur).  v_permlane32_swap(Bprev, Bcur) leaves [Bprev.lo | Bcur.lo] and [Bprev.hi | Bcur.hi]: 8 cycles
        // against a ds_bpermute's 24 (profiles/r02_valu_issue_rates.txt)
        const auto sw32 = __builtin_amdgcn_permlane32_swap(Bprev, Bcur, false, false);
        const uint32_t above = bitop3<0xE4>((uint32_t)sw32[0], (uint32_t)sw32[1], hmask);     // half ? [.. | Bcur.lo] : [Bprev.hi | ..]
        // left / right neighbours by lane shifts (DPP wave_shr / wave_shl, 4 cycles each): the lanes at the ends of a quad row
        // (0 | 32, 31 | 63) are masked anyway
        const uint32_t above_l = dpp0<0x138, 0xF>(above) | q0m;                 // lane x reads x - 1
        const uint32_t above_r = dpp0<0x130, 0xF>(above) | q31m;                // lane x reads x + 1
        const uint32_t rho_l = dpp0<0x138, 0xF>(rho) & ~q0m;
        // max exponent of {w, n0, n1, e} = 32 - min of their leading-zero counts
        const uint32_t Y = __builtin_amdgcn_perm(above_l, above_r, 0x07060100u);   // lo16 = e, hi16 = w
        const u16x2 pm = __builtin_elementwise_min(__builtin_bit_cast(u16x2, above), __builtin_bit_cast(u16x2, Y));
        const uint32_t m4 = min((uint32_t)pm.x, (uint32_t)pm.y);
        const int kap_e = 31 - (int)m4;                             // max_e - 1 (negative when all insignificant)
        // kappa = max(1, max_e - 1) when the quad has two or more significant samples, else 1: bit rho of 0xFEE8 = popcount(rho) >= 2
        const uint32_t gmask = sign_mask(0xFEE80000u << (15u - rho));
        const uint32_t kappa = (uint32_t)max(1, kap_e & (int)gmask);
        // byte offset of the quad's entry in the VLC table (4-byte entries): eps << 2 | rho << 6 | ctx << 10 | other rows << 13,
        // ctx = n_w | rl << 1 | n_e << 2 with n_w / n_e = "both upper neighbours insignificant" (bit 7 of the packed counts)
        uint32_t coff = bitop3<0xEA>(((above_l >> 16) & above) << 3, 0x400u, 0x2000u);
        coff = bitop3<0xEA>(((above >> 16) & above_r) << 5, 0x1000u, coff);
        coff = bitop3<0xEA>((rho_l & 0xCu) + 0x7FCu, 0x800u, coff);
        bool cq0 = (coff & 0x1C00u) == 0x1400u;                         // c_q == 0
        if (it == 0) {                                                  // first quad row (:652, :709)
            const uint32_t cq_first = (rho_l >> 1) | (rho_l & 1u);
            coff = half ? coff : cq_first << 10;
            cq0 = half ? cq0 : cq_first == 0u;
        }
        const uint32_t U = max(emax, kappa);
        const uint32_t u = U - kappa;
        // eps: the samples that attain the maximum exponent (byte of C == cm), only if u > 0.  Byte-wise C - cm is 0 exactly
        // there (no borrows: every byte >= cm); (x & 0x1F) + 0x7F keeps bit 7 clear for a zero byte only.  (Insignificant
        // samples may pass as well: rho masks them out.)
        const uint32_t X = C - __builtin_amdgcn_perm(cm, cm, 0u);             // cm in every byte
        const uint32_t Yz = (X & 0x1F1F1F1Fu) + 0x7F7F7F7Fu;
        const uint32_t e4 = __builtin_amdgcn_udot4(bitop3<0x30>(0x01010101u, Yz >> 7, 0u), 0x20100804u, 0u, false);      // eps << 2
        const uint32_t um = sign_mask(0u - u);
        const uint32_t off = ((rho2 << 4) | coff) | bitop3<0x80>(e4, rho2, um);
        o.tuple = *reinterpret_cast<const uint32_t*>(vtab + off);
        const uint32_t qy = 2 * it + half;
        const bool active = FULL || (qx < QW && qy < QH);
        o.H = __ballot(active && cq0);
        o.V = o.H & __ballot(rho != 0);
        o.R = N; o.U = U; o.u = u;
        Bprev = Bcur;
        // (in the FULL loop unconditional: a refill that only some paths issue leaves the compiler's wait for the table entry no
        //  load in flight it may count on, and it waits for all of them)
        if constexpr (decltype(refill_c)::value) {
            if constexpr (FULL) {
                __builtin_amdgcn_sched_barrier(0);         // (behind the table load, on every path: the same count of loads in flight)
                fetch(it + 4u, nbuf);
            } else if (it + 4u < itn) fetch(it + 4u, nbuf);
        }
    };
    auto stage1 = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[4]) { stage1r(it, o, nbuf, std::true_type{}); };
    auto stage1_last = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[4]) { stage1r(it, o, nbuf, std::false_type{}); };   // nothing left to fetch

    auto stage2 = [&](uint32_t it, const Stage1& s) {
        const uint32_t qy = 2 * it + half;
        const bool active = FULL || (qx < QW && qy < QH);
        const uint32_t U = s.U, u = s.u;
        uint32_t tuple = s.tuple;
        if constexpr (!FULL) tuple = active ? tuple : 0u;

        // ---- MagSgn: bit counts m = U - e_k for significant samples, two at a time: M02 = m0 | m2 << 16, M13 = m1 | m3 << 16
        //      (the table entry carries e_k spread the same way)
        const uint32_t U2 = __builtin_amdgcn_perm(U, U, 0x05040100u);
        const uint32_t M02 = pk_mul_lo_u16(s.R & 0x00010001u, U2) - (tuple & 0x00010001u);
        const uint32_t M13 = pk_mul_lo_u16((s.R >> 8) & 0x00010001u, U2) - ((tuple >> 8) & 0x00010001u);
        const uint32_t Msum = M02 + M13;
        const uint32_t m01 = Msum & 0xFFFFu, m23 = Msum >> 16;
        const uint32_t ms_len = m01 + m23;
        const uint32_t m2 = M02 >> 16, m3 = M13 >> 16;                  // (v_bfe / shifts take m0, m1 from the low 5 bits of M02, M13)
        uint32_t vm[4];
        if constexpr (PK) {
            vm[0] = __builtin_amdgcn_ubfe(s.vv[0], 0u, M02); vm[2] = __builtin_amdgcn_ubfe(s.vv[0], 16u, m2);
            vm[1] = __builtin_amdgcn_ubfe(s.vv[1], 0u, M13); vm[3] = __builtin_amdgcn_ubfe(s.vv[1], 16u, m3);
        } else {
            vm[0] = __builtin_amdgcn_ubfe(s.vv[0], 0u, M02); vm[2] = __builtin_amdgcn_ubfe(s.vv[2], 0u, m2);
            vm[1] = __builtin_amdgcn_ubfe(s.vv[1], 0u, M13); vm[3] = __builtin_amdgcn_ubfe(s.vv[3], 0u, m3);
        }

        // ---- VLC + UVLC: every lane places the bits of its own quad inside the pair's window ----
        uint32_t ui = u;
        bool xev = false, xv = false;
        if (it == 0) {
            const uint32_t up = quad_swap(u);
            const bool first = half == 0;
            const bool both = first && u > 2 && up > 2;
            ui = both ? u - 2 : u;
            ui = (first && odd && up > 2 && u > 0 && u <= 2) ? 32u + u : ui;
            xev = first && odd && u > 0 && up > 0;
            xv = xev && min(u, up) > 2;
        }
        const uint2 ue = uvlc_l[ui];
        const uint32_t A = (tuple >> 25) | ue.x;                      // cwd | pre<<8 | suf<<16
        const uint32_t Lw = ((tuple >> 4) & 7u) | ue.y;               // len | pl<<8 | sl<<16
        const uint32_t Lp = quad_swap(Lw);
        const uint32_t S = Lw + Lp;
        const uint32_t Lq = odd ? Lp : 0u;
        const uint32_t slen = S & 0xFFu, spl = (S >> 8) & 0xFFu;
        const uint32_t off_pre = slen + ((Lq >> 8) & 0xFFu);
        const uint32_t off_suf = slen + spl + (Lq >> 16);
        const uint32_t cl = slen + spl + (S >> 16);                   // bits of the whole pair
        const uint32_t wv = ((A & 0xFFu) << (Lq & 0xFFu)) | (((A >> 8) & 0xFFu) << off_pre) | ((A >> 16) << off_suf);

        // ---- one packed prefix sum: low 16 bits MagSgn, high 16 bits VLC (pair total on even lanes)
        const uint32_t packed = ms_len | ((odd ? 0u : cl) << 16);
        const uint32_t incl = wave_incl_scan(packed);
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t mpos = ms_bits + ((incl - packed) & 0xFFFFu);
        const uint32_t vpos = vlc_bits + (incl >> 16) - cl;
        ms_bits += tot & 0xFFFFu;
        vlc_bits += tot >> 16;

        // The raw streams have room for what real content needs, not for the worst case (the LDS per wave is what limits
        // the occupancy: 16 -> 24 waves per CU is 15-18 % of this kernel): a block that outgrows them stops writing and is
        // coded again by the fallback launch with worst-case buffers (wave-uniform test, never taken on natural images).
        lds_full = lds_full || ms_bits > L.ms_cap_bits || vlc_bits > L.vlc_cap_bits;
        if (lds_full) return;
        // `narrow`: the exponent bounds every quad's four values to 64 bits.  Deep content (16-bit, quantised) is not bounded that
        // way but mostly is that small: one wave-uniform test of the pair sums takes the one-piece path whenever every lane's fit
        // Only the lanes that have bits to put take part in the LDS atomics: a quad without MagSgn bits (nothing significant in it)
        // sits at the SAME bit position as its neighbours, and 64 atomic ORs on one dword are served one after the other -- on
        // flat content (every quad empty) that made K3 0.68 ms per 8K frame against 0.27 for real content, the waves waiting on
        // LDS (SQ_LDS_ADDR_CONFLICT 198 M cycles per frame; profiles/r04_small_frames.txt).  Costs the dense case one compare each: K3 +0.7 % on the headline
        // frame, A/B on one box.
        if ((packed & 0xFFFFu) != 0u) {
            if (narrow || !__ballot(max(m01, m23) > 32u)) {
                const uint32_t v01 = vm[0] | (vm[1] << (M02 & 31u));
                const uint32_t v23 = vm[2] | (vm[3] << m2);
                or_bits64(ms_raw, mpos, (uint64_t)v01 | ((uint64_t)v23 << m01), ms_len);
            } else {
                or_bits64(ms_raw, mpos, (uint64_t)vm[0] | ((uint64_t)vm[1] << (M02 & 0xFFFFu)), m01);
                or_bits64(ms_raw, mpos + m01, (uint64_t)vm[2] | ((uint64_t)vm[3] << m2), m23);
            }
        }
        if (cl != 0u) or_bits32(vlc_raw, vpos, wv);

        // ---- MEL events (wave-uniform, scalar unit) ------------------------------------------------
        const uint64_t H = s.H, V = s.V;
        uint64_t Hm = H;
        if (it != 0 && !Hm) return;
        if (it == 0) {
            // the first quad row's events, pair by pair: quad 2p's, quad 2p + 1's, the pair's u-event (:652-730).  Laid out in that
            // order over lanes 0 .. 47 (lane 3p + j takes its flag from lane 2p or 2p + 1: one ds_bpermute) and coded by the loop
            // below like every other row's -- walked pair by pair on the scalar unit they cost ~1 000 scalar instructions per block
            // on dense content, where the u-event of all 16 pairs is there (r05, profiles/r05_k3_pairs.txt)
            const uint32_t lo = (uint32_t)lane & 31u;
            const uint32_t fl = (((uint32_t)H >> lo) & 1u) | (xev ? 2u : 0u) | ((((uint32_t)V >> lo) & 1u) << 2) | (xv ? 8u : 0u);
            const uint32_t third = ((uint32_t)lane * 43u) >> 7, j = (uint32_t)lane - 3u * third;      // lane / 3, lane % 3
            const uint32_t src = 2u * third + (j ? 1u : 0u);
            const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * src), (int)fl) >> (j == 2u ? 1u : 0u);
            const uint64_t EH = __ballot(lane < 48 && (g & 1u));
            const uint64_t EV = EH & __ballot((g & 4u) != 0u);
            mel_first_row(melp, EH, EV);
            Hm &= 0xFFFFFFFF00000000ull;
        }
        // the events in quad order: the zero events up to the next significant quad as one run, then that quad's event
        if (Hm) mel_row(melp, Hm, V);
    };

    Stage1 sE, sO;                       // unrolled by two so that no pipeline register is ever copied
    stage1(0, sE, n0);
    if constexpr (FULL) {
        uint32_t it = 0;
        for (; it + 8 < itn; it += 4) {                    // (the back edge always follows a stage 1: one path, one count of loads in flight)
            stage1(it + 1, sO, n1); stage2(it, sE);
            stage1(it + 2, sE, n2); stage2(it + 1, sO);
            stage1(it + 3, sO, n3); stage2(it + 2, sE);
            stage1(it + 4, sE, n0); stage2(it + 3, sO);
        }
        // the last eight iterations: the sets of the last four are not refilled -- a load nobody reads still has to land before
        // its register is written again, and phase B would begin with a wait for four of them
        stage1(it + 1, sO, n1); stage2(it, sE);
        stage1(it + 2, sE, n2); stage2(it + 1, sO);
        stage1(it + 3, sO, n3); stage2(it + 2, sE);
        stage1_last(it + 4, sE, n0); stage2(it + 3, sO);
        stage1_last(it + 5, sO, n1); stage2(it + 4, sE);
        stage1_last(it + 6, sE, n2); stage2(it + 5, sO);
        stage1_last(it + 7, sO, n3); stage2(it + 6, sE);
        stage2(it + 7, sO);
    } else {
        for (uint32_t it = 0; it < itn; it += 4) {
            if (it + 1 < itn) stage1(it + 1, sO, n1);
            stage2(it, sE);
            if (it + 1 < itn) { if (it + 2 < itn) stage1(it + 2, sE, n2); stage2(it + 1, sO); }
            if (it + 2 < itn) { if (it + 3 < itn) stage1(it + 3, sO, n3); stage2(it + 2, sE); }
            if (it + 3 < itn) { if (it + 4 < itn) stage1(it + 4, sE, n0); stage2(it + 3, sO); }
        }
        // (a use of every set: the compiler's bookkeeping of loads in flight is path-insensitive, and what it cannot prove consumed
        //  here would cost the code behind phase A -- shared with the FULL blocks -- a wait for everything at its first register reuse)
        asm volatile("" : : "v"(n0[0]), "v"(n0[1]), "v"(n0[2]), "v"(n0[3]), "v"(n1[0]), "v"(n1[1]), "v"(n1[2]), "v"(n1[3]),
                            "v"(n2[0]), "v"(n2[1]), "v"(n2[2]), "v"(n2[3]), "v"(n3[0]), "v"(n3[1]), "v"(n3[2]), "v"(n3[3]));
    }
    };   // phase_a

    // ---- Phase A of a FULL block with TWO quads per lane (r05) ------------------------------------------------------------
    // Lane l: quad row r = l >> 4 of the iteration's FOUR, quads 2c and 2c + 1 (c = l & 15) -- the two quads of a VLC pair,
    // neighbours in the MagSgn stream.  What a wave instruction costs does not depend on how many quads a lane holds, so
    // everything that is per PAIR or per wave step is paid once for 128 quads instead of once for 64: one prefix sum, one
    // position, one pair of LDS atomics for the pair's MagSgn bits (<= 64 bits on all but deep content) and one for its VLC
    // window, the pair's VLC layout without a lane exchange, half the MEL ballots' scalar overhead; 8 iterations per block.
    // The exponents of the row above come from the lanes 16 below (v_permlane16_swap + v_permlane32_swap: the four bottom-row
    // exponents of a lane's two quads travel as the bytes of ONE register), the quads left and right by DPP lane shifts.
    auto phase_a2 = [&]() {
        constexpr bool PK = H16;
        constexpr int NW = PK ? 4 : 8;                      // dwords of a sample set: PK [0] A top, [1] A bottom, [2] B top, [3] B bottom
        const uint32_t rr = (uint32_t)lane >> 4, cc = (uint32_t)lane & 15u;
        const uint32_t c0m = cc == 0 ? 0xFFFFFFFFu : 0u, c15m = cc == 15 ? 0xFFFFFFFFu : 0u;
        const uint32_t oddm = (rr & 1u) ? 0xFFFFFFFFu : 0u, row0m = rr == 0 ? 0xFFFFFFFFu : 0u;
        const uint32_t lane_off2 = 2u * rr * stride_b + 4u * cc * EB;
        typedef int i32x4 __attribute__((ext_vector_type(4)));

        auto fetch2 = [&](uint32_t it, int32_t (&r)[NW]) {
            const uint32_t o0 = lane_off2 + it * (8u * stride_b), o1 = o0 + stride_b;
            if constexpr (PK) {
                const i32x2 q0 = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(srcb + o0));
                const i32x2 q1 = __builtin_nontemporal_load(reinterpret_cast<const i32x2*>(srcb + o1));
                r[0] = q0.x; r[1] = q1.x; r[2] = q0.y; r[3] = q1.y;
            } else {       // quad A: [0] (x0,y0) [1] (x0,y0+1) [2] (x0+1,y0) [3] (x0+1,y0+1); quad B: [4..7]
                const i32x4 q0 = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(srcb + o0));
                const i32x4 q1 = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(srcb + o1));
                r[0] = q0.x; r[2] = q0.y; r[1] = q1.x; r[3] = q1.y;
                r[4] = q0.z; r[6] = q0.w; r[5] = q1.z; r[7] = q1.w;
            }
        };
        int32_t n0[NW], n1[NW], n2[NW], n3[NW];
        fetch2(0, n0); fetch2(1, n1); fetch2(2, n2); fetch2(3, n3);

        struct Stage1 {
            uint32_t vv[2][PK ? 2 : 4];
            uint32_t SM[2], U[2], u[2], tuple[2];      // SM: 0xFF in the byte of every significant sample
            uint2 ue[2];                               // the UVLC entries of u (rows behind the first: looked up a stage ahead, so that
                                                       // stage 2 does not begin with the LDS round trip)
            uint64_t H[2], V[2];
        };
        uint32_t Aprev = 0xFFFFFFFFu;          // (R1, R1, R3, R3) of the iteration before: its row 3 is this iteration's row above row 0

        // one quad's samples -> exponents (leading-zero form, one per byte) and MagSgn values (as in phase_a)
        auto analyse = [&](const int32_t* sm, uint32_t (&vv)[PK ? 2 : 4], uint32_t& C) {
            if constexpr (PK) {
                const int32_t nw0 = sm[0], nw1 = sm[1];
                const i16x2 w0 = __builtin_bit_cast(i16x2, nw0), w1 = __builtin_bit_cast(i16x2, nw1);
                const u16x2 p0 = __builtin_bit_cast(u16x2, __builtin_elementwise_max(w0, -w0));
                const u16x2 p1 = __builtin_bit_cast(u16x2, __builtin_elementwise_max(w1, -w1));
                or3_now(ovf, __builtin_bit_cast(uint32_t, p0), __builtin_bit_cast(uint32_t, p1));
                const u16x2 one2 = {1, 1};
                const uint32_t t0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(p0 + p0, one2));
                const uint32_t t1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(p1 + p1, one2));
                ffbh_u16_to_byte<0, 0>(C, t0); ffbh_u16_to_byte<1, 0>(C, t1);
                ffbh_u16_to_byte<2, 1>(C, t0); ffbh_u16_to_byte<3, 1>(C, t1);
                vv[0] = pk_sub_u16(t0, pk_lshr15_u16(~(uint32_t)nw0));
                vv[1] = pk_sub_u16(t1, pk_lshr15_u16(~(uint32_t)nw1));
            } else {
                uint32_t t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t mag;
                    if constexpr (IRREV) {
                        const float cf = __int_as_float(sm[i]);
                        const float q = __fmul_rn(fabsf(cf), inv_step);
                        mag = min((uint32_t)q, lim);
                    } else {
                        mag = (uint32_t)max(sm[i], -sm[i]);
                    }
                    ovf |= mag;
                    t[i] = (mag << 1) - 1u;
                    vv[i] = t[i] - 1u + ((uint32_t)sm[i] >> 31);
                }
                ffbh_i32_to_byte<0>(C, t[0]); ffbh_i32_to_byte<1>(C, t[1]);
                ffbh_i32_to_byte<2>(C, t[2]); ffbh_i32_to_byte<3>(C, t[3]);
            }
        };

        auto stage1r = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[NW], auto refill_c) {
            uint32_t C[2];
            analyse(nbuf, o.vv[0], C[0]);
            analyse(nbuf + NW / 2, o.vv[1], C[1]);
            uint32_t N[2], rho[2], rho2[2], cm[2], emax[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                N[q] = bitop3<0x30>(0x01010101u, C[q] >> 7, 0u);
                rho[q] = __builtin_amdgcn_udot4(N[q], 0x08040201u, 0u, false);
                rho2[q] = __builtin_amdgcn_udot4(N[q], 0x20100804u, 0u, false);
                cm[q] = min_of_bytes(C[q]);
                emax[q] = 32u - min(cm[q], 32u);
            }
            // ---- the row above: the bottom-row exponents of sample columns 4c .. 4c + 3 as the four bytes of one register
            const uint32_t Bp = __builtin_amdgcn_perm(C[1], C[0], 0x07050301u);      // A byte 1, A byte 3, B byte 1, B byte 3
            // rows (R0 R1 R2 R3) -> (P3 R0 R1 R2), P = the iteration before:
            //   permlane16_swap(X, X)        leaves (R0 R0 R2 R2) and (R1 R1 R3 R3)
            //   permlane32_swap(Aprev, that) leaves (P1 P1 R1 R1) and (P3 P3 R3 R3)
            const auto s16 = __builtin_amdgcn_permlane16_swap(Bp, Bp, false, false);
            const uint32_t A1 = (uint32_t)s16[1];
            const auto s32 = __builtin_amdgcn_permlane32_swap(Aprev, A1, false, false);
            const uint32_t tsel = bitop3<0xE4>((uint32_t)s32[1], (uint32_t)s32[0], row0m);      // row 0 ? P3 : (row 2) R1
            const uint32_t Y = bitop3<0xE4>((uint32_t)s16[0], tsel, oddm);                       // rows 1, 3 ? R0, R2
            Aprev = A1;
            const uint32_t Yl = dpp0<0x138, 0xF>(Y) | c0m;                      // lane x reads x - 1
            const uint32_t Yr = dpp0<0x130, 0xF>(Y) | c15m;                     // lane x reads x + 1
            const uint32_t rhoBl = dpp0<0x138, 0xF>(rho[1]) & ~c0m;             // the quad left of quad A
            // (w, n0, n1, e) of the two quads: bytes of one register each
            const uint32_t W[2] = {__builtin_amdgcn_alignbit(Y, Yl, 24), __builtin_amdgcn_alignbit(Yr, Y, 8)};
            const uint32_t rl[2] = {rhoBl, rho[0]};
            bool cq0[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t m4 = min_of_bytes(W[q]);
                const int kap_e = 31 - (int)m4;
                const uint32_t gmask = sign_mask(0xFEE80000u << (15u - rho[q]));
                const uint32_t kappa = (uint32_t)max(1, kap_e & (int)gmask);
                const uint32_t F = W[q] & (W[q] >> 8);         // bit 7: w & n0 both insignificant, bit 23: n1 & e
                uint32_t coff = bitop3<0xEA>(F << 3, 0x400u, 0x2000u);
                coff = bitop3<0xEA>(F >> 11, 0x1000u, coff);
                coff = bitop3<0xEA>((rl[q] & 0xCu) + 0x7FCu, 0x800u, coff);
                cq0[q] = (coff & 0x1C00u) == 0x1400u;
                if (it == 0) {                                                  // first quad row of the block: lanes 0 .. 15
                    const uint32_t cq_first = (rl[q] >> 1) | (rl[q] & 1u);
                    coff = rr ? coff : cq_first << 10;
                    cq0[q] = rr ? cq0[q] : cq_first == 0u;
                }
                const uint32_t U = max(emax[q], kappa);
                const uint32_t u = U - kappa;
                const uint32_t X = C[q] - __builtin_amdgcn_perm(cm[q], cm[q], 0u);
                const uint32_t Yz = (X & 0x1F1F1F1Fu) + 0x7F7F7F7Fu;
                const uint32_t e4 = __builtin_amdgcn_udot4(bitop3<0x30>(0x01010101u, Yz >> 7, 0u), 0x20100804u, 0u, false);
                const uint32_t um = sign_mask(0u - u);
                const uint32_t off = ((rho2[q] << 4) | coff) | bitop3<0x80>(e4, rho2[q], um);
                o.tuple[q] = *reinterpret_cast<const uint32_t*>(vtab + off);
                o.SM[q] = N[q] * 255u; o.U[q] = U; o.u[q] = u;
                if (it != 0) o.ue[q] = uvlc_l[u];
            }
            o.H[0] = __ballot(cq0[0]); o.V[0] = o.H[0] & __ballot(rho[0] != 0);
            o.H[1] = __ballot(cq0[1]); o.V[1] = o.H[1] & __ballot(rho[1] != 0);
            if constexpr (decltype(refill_c)::value) {
                __builtin_amdgcn_sched_barrier(0);
                fetch2(it + 4u, nbuf);
            }
        };
        auto stage1 = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[NW]) { stage1r(it, o, nbuf, std::true_type{}); };
        auto stage1_last = [&](uint32_t it, Stage1& o, int32_t (&nbuf)[NW]) { stage1r(it, o, nbuf, std::false_type{}); };

        auto stage2 = [&](uint32_t it, const Stage1& s) {
            // ---- MagSgn bit counts m_i = U - e_k,i of the significant samples, one per BYTE (sample i in byte i: the table entry
            //      carries e_k spread that way), and the values cut to them
            uint32_t vm[2][4], M4[2], m1[2], m2[2], m3[2], m01[2], m23[2], qlen[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                M4[q] = (__builtin_amdgcn_perm(s.U[q], s.U[q], 0u) - (s.tuple[q] & 0x01010101u)) & s.SM[q];
                qlen[q] = __builtin_amdgcn_udot4(M4[q], 0x01010101u, 0u, false);
                m01[q] = __builtin_amdgcn_udot4(M4[q], 0x00000101u, 0u, false);
                m23[q] = qlen[q] - m01[q];
                m1[q] = M4[q] >> 8; m2[q] = M4[q] >> 16; m3[q] = M4[q] >> 24;       // (bit-field widths and shift counts take the low 5 bits)
                if constexpr (PK) {
                    vm[q][0] = __builtin_amdgcn_ubfe(s.vv[q][0], 0u, M4[q]); vm[q][2] = __builtin_amdgcn_ubfe(s.vv[q][0], 16u, m2[q]);
                    vm[q][1] = __builtin_amdgcn_ubfe(s.vv[q][1], 0u, m1[q]); vm[q][3] = __builtin_amdgcn_ubfe(s.vv[q][1], 16u, m3[q]);
                } else {
                    vm[q][0] = __builtin_amdgcn_ubfe(s.vv[q][0], 0u, M4[q]); vm[q][2] = __builtin_amdgcn_ubfe(s.vv[q][2], 0u, m2[q]);
                    vm[q][1] = __builtin_amdgcn_ubfe(s.vv[q][1], 0u, m1[q]); vm[q][3] = __builtin_amdgcn_ubfe(s.vv[q][3], 0u, m3[q]);
                }
            }
            const uint32_t ms_len = qlen[0] + qlen[1];

            // ---- VLC + UVLC of the pair: cwd A | cwd B | prefix A | prefix B | suffix A | suffix B
            uint32_t uiA = s.u[0], uiB = s.u[1];
            bool xev = false, xv = false;
            if (it == 0) {                                                      // first quad row (:652, :709): lanes 0 .. 15
                const bool first = rr == 0;
                const uint32_t uA = s.u[0], uB = s.u[1];
                const bool both = first && uA > 2 && uB > 2;
                uiA = both ? uA - 2 : uA;
                uiB = both ? uB - 2 : uB;
                uiB = (first && uA > 2 && uB > 0 && uB <= 2) ? 32u + uB : uiB;
                xev = first && uA > 0 && uB > 0;
                xv = xev && min(uA, uB) > 2;
            }
            const uint2 ueA = it == 0 ? uvlc_l[uiA] : s.ue[0], ueB = it == 0 ? uvlc_l[uiB] : s.ue[1];
            const uint32_t AA = (s.tuple[0] >> 25) | ueA.x, AB = (s.tuple[1] >> 25) | ueB.x;       // cwd | pre << 8 | suf << 16
            const uint32_t LA = ((s.tuple[0] >> 4) & 7u) | ueA.y, LB = ((s.tuple[1] >> 4) & 7u) | ueB.y;   // len | pl << 8 | sl << 16
            const uint32_t S = LA + LB;
            const uint32_t lenA = LA & 0xFFu, slen = S & 0xFFu;
            const uint32_t o_preB = slen + ((LA >> 8) & 0xFFu);
            const uint32_t o_sufA = slen + ((S >> 8) & 0xFFu);
            const uint32_t o_sufB = o_sufA + (LA >> 16);
            const uint32_t cl = o_sufA + (S >> 16);                             // bits of the whole pair
            const uint32_t wv = (AA & 0xFFu) | ((AB & 0xFFu) << lenA) | (((AA >> 8) & 0xFFu) << slen) | (((AB >> 8) & 0xFFu) << o_preB) |
                                ((AA >> 16) << o_sufA) | ((AB >> 16) << o_sufB);

            // ---- one packed prefix sum: low 16 bits MagSgn, high 16 bits VLC
            const uint32_t packed = ms_len | (cl << 16);
            const uint32_t incl = wave_incl_scan(packed);
            const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t mpos = ms_bits + ((incl - packed) & 0xFFFFu);
            const uint32_t vpos = vlc_bits + (incl >> 16) - cl;
            ms_bits += tot & 0xFFFFu;
            vlc_bits += tot >> 16;
            // (one test for both streams: a capacity minus a count that outgrew it has its top bit set)
            if (((L.ms_cap_bits - ms_bits) | (L.vlc_cap_bits - vlc_bits)) >> 31) { lds_full = true; return; }

            if (ms_len != 0u) {
                if (!__ballot(max(qlen[0], qlen[1]) > 32u)) {
                    // both quads fit a word each (8-bit content: ~13 bits per quad): the pair goes out in one piece
                    const uint32_t qa = (vm[0][0] | (vm[0][1] << (M4[0] & 31u))) | ((vm[0][2] | (vm[0][3] << (m2[0] & 31u))) << (m01[0] & 31u));
                    const uint32_t qb = (vm[1][0] | (vm[1][1] << (M4[1] & 31u))) | ((vm[1][2] | (vm[1][3] << (m2[1] & 31u))) << (m01[1] & 31u));
                    or_bits64(ms_raw, mpos, (uint64_t)qa | ((uint64_t)qb << qlen[0]), ms_len);
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const uint32_t at = q ? mpos + qlen[0] : mpos;
                        if (narrow || !__ballot(max(m01[q], m23[q]) > 32u)) {
                            const uint32_t v01 = vm[q][0] | (vm[q][1] << (M4[q] & 31u));
                            const uint32_t v23 = vm[q][2] | (vm[q][3] << (m2[q] & 31u));
                            or_bits64(ms_raw, at, (uint64_t)v01 | ((uint64_t)v23 << m01[q]), qlen[q]);
                        } else {
                            or_bits64(ms_raw, at, (uint64_t)vm[q][0] | ((uint64_t)vm[q][1] << (M4[q] & 0xFFu)), m01[q]);
                            or_bits64(ms_raw, at + m01[q], (uint64_t)vm[q][2] | ((uint64_t)vm[q][3] << (m2[q] & 0xFFu)), m23[q]);
                        }
                    }
                }
            }
            if (cl != 0u) or_bits32(vlc_raw, vpos, wv);

            // ---- MEL events (wave-uniform, scalar unit): in quad order = lane by lane, quad A before quad B
            uint64_t HA = s.H[0], VA = s.V[0], HB = s.H[1], VB = s.V[1];
            if (it != 0 && !(HA | HB)) return;
            if (it == 0) {
                // The first quad row's events come pair by pair: quad A's, quad B's, the pair's u-event (:652-730) -- on dense content
                // the u-event of all 16 pairs.  Walked pair by pair on the scalar unit (three tests and up to three trips through the
                // state machine per pair) they were ~1 000 of K3's 2 300 scalar instructions per block, 15 % of its time for ~17 events
                // (profiles/r05_k3_pairs.txt).  Now the 48 flags are laid out in event order over lanes 0 .. 47 -- lane 3p + j takes
                // flag j of pair p from lane p (one ds_bpermute) -- and ONE ballot pair feeds the same run-skipping loop as every other row.
                const uint32_t lo = (uint32_t)lane & 31u;
                const uint32_t fl = (((uint32_t)HA >> lo) & 1u) | ((((uint32_t)HB >> lo) & 1u) << 1) | (xev ? 4u : 0u) |
                                    ((((uint32_t)VA >> lo) & 1u) << 3) | ((((uint32_t)VB >> lo) & 1u) << 4) | (xv ? 32u : 0u);
                const uint32_t third = ((uint32_t)lane * 43u) >> 7;                     // lane / 3
                const uint32_t g = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * third), (int)fl) >> ((uint32_t)lane - 3u * third);
                const bool in_row = lane < 48;
                const uint64_t EH = __ballot(in_row && (g & 1u)), EV = EH & __ballot((g & 8u) != 0u);
                mel_first_row(melp, EH, EV);
                HA &= ~0xFFFFull; HB &= ~0xFFFFull;
            }
            if (!(HA | HB)) return;
            MelState mel = mel_unpack(melp);                       // (8-bit content with empty quads: the run-skipping loop on the whole state)
            while (HA | HB) {
                const uint64_t oa = HA & VA, ob = HB & VB;
                if (!(oa | ob)) {
                    mel_zero_run(mel, mel_buf, (uint32_t)(__builtin_popcountll(HA) + __builtin_popcountll(HB)), lane == 0);
                    break;
                }
                const uint32_t la = oa ? (uint32_t)__builtin_ctzll(oa) : 64u, lb = ob ? (uint32_t)__builtin_ctzll(ob) : 64u;
                const bool a_first = (int)(la - lb) <= 0;          // (the same lane: quad A comes first)
                const uint32_t ln = a_first ? la : lb;
                const uint64_t bit = 1ull << ln, below = bit - 1;
                // everything of the lanes below, and quad A of this lane when the event is quad B's
                const uint64_t ma = a_first ? below : (below | bit);
                mel_zero_run(mel, mel_buf, (uint32_t)(__builtin_popcountll(HA & ma) + __builtin_popcountll(HB & below)), lane == 0);
                mel_event(mel, mel_buf, 1, lane == 0);
                HA &= ~(below | bit);
                HB &= a_first ? ~below : ~(below | bit);
            }
            mel_pack(melp, mel);
        };

        Stage1 sE, sO;
        stage1(0, sE, n0);
        stage1(1, sO, n1); stage2(0, sE);
        stage1(2, sE, n2); stage2(1, sO);
        stage1(3, sO, n3); stage2(2, sE);
        stage1_last(4, sE, n0); stage2(3, sO);
        stage1_last(5, sO, n1); stage2(4, sE);
        stage1_last(6, sE, n2); stage2(5, sO);
        stage1_last(7, sO, n3); stage2(6, sE);
        stage2(7, sO);
    };   // phase_a2
    // The pair form for packed (8-bit reversible) content only: with four 32-bit samples per quad it needs 110 - 141 registers, four or
    // three waves per SIMD, and codes cfg3 13 % SLOWER than the one-quad form (same box: 0.537 against 0.476 ms).
    // (8-byte row loads: block origin and stride multiples of four samples -- every 64 x 64 block of a Mallat plane)
    if constexpr (H16) {
        if (full && ((bd.px | a.stride) & 3u) == 0) phase_a2(); else phase_a(std::false_type{});
    } else {
        if (full) phase_a(std::true_type{}); else phase_a(std::false_type{});
    }
    if (lds_full) {                       // hand the block to the fallback launch (kernels.h: HtArgs::ovf_list)
        if (lane == 0) {
            const unsigned long long at = __hip_atomic_fetch_add(a.alloc + 2 + class_id, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.ovf_list[a.ovf_base + at] = tile * a.sel_count + li;
        }
        return;
    }
    // magnitudes beyond Kmax+1 bits: outside the contract (see header) -> flag, host reports it
    if constexpr (H16) ovf = (ovf & 0xFFFFu) | (ovf >> 16);          // FULL blocks accumulate two magnitudes per word
    if (!IRREV && __ballot((ovf >> (kmax + 1)) != 0)) {
        if (lane == 0) atomicOr(reinterpret_cast<unsigned int*>(a.alloc), 2u);
    }
    __syncthreads();

    // ---- reserve the block's bytes in the arena now, from an upper bound of its length, so that the
    //      round trip of the device-scope atomic (it executes at the memory side) hides behind phase B.
    //      Stuffing adds at most one bit per 15 raw bits; MEL grows by at most 2 more bytes.
    MelState mel = mel_unpack(melp);
    const uint32_t len_ub = (ms_bits + ms_bits / 15u) / 8u + (vlc_bits + vlc_bits / 15u) / 8u + mel.pos + 8u;
    unsigned long long base_off = 0;
    if (lane == 0) base_off = arena_alloc(a.alloc, gid & a.region_mask, len_ub, a.chunk_units);

    // ================= phase B: stuffing, termination, emission (oracle/ht_wave_model.c, orc_ht_model_phase_b2) =====
    // Byte stuffing only moves byte boundaries after rare events (MagSgn: the byte after a 0xFF has 7 bits; VLC: a byte after
    // one > 0x8F whose low 7 bits are ones has 7).  "Speculative windows": 64 lanes cut 4 output bytes each out of the raw
    // stream as if no event fell into the window and test their own bytes; without an event (one ballot) the 256 bytes are
    // final and stored as they are, otherwise everything before the first event is, the event's 7-bit byte is dealt with on
    // the spot and the next window starts behind it.  r02 found the events with a walker first, kept them in bitmaps with
    // prefix counts, and looked every output dword's start up afterwards (~2.2x the vector instructions of this form).
    // ---- B1: VLC bytes are stored backwards from the block's END, so their number comes first.
    //      r01-r04: a walker that only counts the 7-bit bytes (count_vlc_events), the bytes themselves emitted at the end (B4) --
    //      the stream is looked at twice, and the walker alone was 6 % of K3 (a what-if build without it: 0.2674 -> 0.2509 ms,
    //      -199 vector / -155 scalar instructions per block; profiles/r05_k3_pairs.txt).  r05: the emission runs HERE, forwards,
    //      into a staging area in LDS (vst: byte i of the VLC segment at vst[i]); what the termination needs -- the number of whole
    //      bytes, the bits left over -- falls out of it, and B4 copies the staged bytes out reversed.
    uint8_t* const vst = reinterpret_cast<uint8_t*>(uvlc_l);        // (the UVLC table is done with)
    typedef uint32_t u32_lds_any __attribute__((aligned(1)));
    uint32_t nv, vposr;
    {
        uint32_t s = 0, i = 0, prev = 0xFFu << 24;
        while (true) {
            const uint32_t avail = (vlc_bits - s) >> 3;              // whole 8-bit bytes left if no 7-bit byte comes
            if (avail == 0) break;
            const uint32_t start = s + 32u * (uint32_t)lane;
            const uint32_t sw = start >> 5;
            const uint32_t win = __builtin_amdgcn_alignbit(vlc_raw[sw + 1], vlc_raw[sw], start);
            // the byte before each of the four: lane - 1's top byte (lane 0: the window before, `prev`)
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)prev, (int)win, 0x138, 0xF, 0xF, false);
            const uint32_t pw = __builtin_amdgcn_alignbit(win, before, 24);
            const uint32_t e = (win & 0x7F7F7F7Fu) + 0x01010101u;   // bit 7 of a byte: its low 7 bits are ones
            // flag at bit 4 of a byte: the byte before has bit 7 and one of bits 6..4 (it is > 0x8F), this one's low 7 bits are ones
            uint32_t z = bitop3<0x80>(pw >> 3, (pw >> 2) | (pw >> 1) | pw, e >> 3) & 0x10101010u;
            const uint32_t lane4s = 4u * (uint32_t)lane;
            uint32_t nb = 4;
            if (avail < 256u) {
                nb = min(avail - min(avail, lane4s), 4u);
                z &= nb >= 4u ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
            }
            const uint64_t ballot = __ballot(z != 0);
            const uint32_t F = ballot ? (uint32_t)__ffsll((long long)ballot) - 1u : 64u;
            if ((uint32_t)lane < F) {                                // lanes before the event (all of them without one): their bytes
                if (nb == 4u) *reinterpret_cast<u32_lds_any*>(vst + i + lane4s) = win;
                else {
#pragma unroll 1
                    for (uint32_t k = 0; k < nb; ++k) vst[i + lane4s + k] = (uint8_t)(win >> (8u * k));
                }
            }
            if (!ballot) {
                const uint32_t n = min(avail, 256u);
                // the last byte handed out: what the next window's (or the tail's) first byte follows
                const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)((n - 1u) >> 2));
                prev = (lw >> (8u * ((n - 1u) & 3u))) << 24;
                s += 8u * n; i += n;
                continue;
            }
            const uint32_t zF = (uint32_t)__builtin_amdgcn_readlane((int)z, (int)F);
            const uint32_t winF = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)F);
            const uint32_t b = (uint32_t)(__ffs((int)zF) - 1) >> 3;
            i += 4u * F;
            if ((uint32_t)lane <= b)                                 // the event lane's bytes up to the 7-bit one (0x7F)
                vst[i + lane] = (uint8_t)((winF >> (8u * lane)) & ((uint32_t)lane == b ? 0x7Fu : 0xFFu));
            s += 32u * F + 8u * b + 7u; i += b + 1u; prev = 0x7Fu << 24;
        }
        // a 7-bit byte that takes the stream's last seven bits is a whole byte too
        if (vlc_bits - s == 7u && (prev >> 24) > 0x8Fu &&
            (uint32_t)__builtin_amdgcn_readfirstlane((int)get_bits(vlc_raw, s, 7u)) == 0x7Fu) {
            if (lane == 0) vst[i] = 0x7Fu;
            i += 1u; s += 7u;
        }
        nv = i; vposr = s;
    }
    const uint32_t vused = vlc_bits - vposr;
    const uint32_t vacc = vused ? get_bits(vlc_raw, vposr, vused) : 0;

    // ---- B2: MEL / VLC termination (terminate_mel_vlc :357-385), wave-uniform
    if (mel.run > 0) mel_put_bit(mel, mel_buf, 1, lane == 0);
    uint32_t vextra = 0;
    {
        const int macc = mel.acc << mel.left;
        const int mel_mask = (0xFF << mel.left) & 0xFF;
        const int vlc_mask = 0xFF >> (8 - (int)vused);
        if ((mel_mask | vlc_mask) != 0) {
            const int fuse = macc | (int)vacc;
            if ((((fuse ^ macc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && nv >= 1) {
                mel_put_byte(mel, (uint32_t)fuse & 0xFFu);
            } else {
                mel_put_byte(mel, (uint32_t)macc & 0xFFu);
                vextra = 1;
            }
        }
    }
    mel_flush(mel, mel_buf, lane);
    const uint32_t mel_len = mel.pos;
    const uint32_t vcount = nv + vextra;
    const uint32_t scup = mel_len + vcount + 1;

    // ---- the reserved bytes (16-byte aligned, order of arrival); nothing is written outside them
    base_off = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base_off >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base_off);
    if (base_off + len_ub > a.arena_bytes) {
        if (lane == 0) { a.lengths[gid] = 0; a.offsets[gid] = base_off; atomicOr(reinterpret_cast<unsigned int*>(a.alloc), 1u); }
        return;
    }
    uint8_t* const out = a.arena + base_off;
    typedef uint32_t u32_any __attribute__((aligned(1)));
    const uint32_t lane32 = 32u * (uint32_t)lane, lane4 = 4u * (uint32_t)lane;

    // ---- B3: MagSgn bytes (emit_ms of the model): s = raw bit the next byte starts at, j = its index in the output
    uint32_t ms_len;
    {
        uint32_t s = 0, j = 0;
        bool done = false;
        while (s + 8u <= ms_bits) {
            const uint32_t avail = (ms_bits - s) >> 3;               // whole bytes left if no event comes
            const uint32_t start = s + lane32;
            const uint32_t sw = start >> 5;
            const uint32_t win = __builtin_amdgcn_alignbit(ms_raw[sw + 1], ms_raw[sw], start);    // (the shift takes the low 5 bits)
            // a byte of win is 0xFF <=> that byte of ~win is zero: (x - 0x01010101) & ~x & 0x80808080, exact for the LOWEST flag
            uint32_t z = bitop3<0x80>(~win - 0x01010101u, win, 0x80808080u);
            uint32_t nb = 4;
            if (avail < 256u) {                                      // last window: lanes beyond the stream, a partial lane
                nb = min(avail - min(avail, lane4), 4u);
                z &= nb >= 4u ? 0xFFFFFFFFu : (1u << (8u * nb)) - 1u;
            }
            const uint64_t ballot = __ballot(z != 0);
            if (!ballot) {
                if (avail >= 256u) {                                 // (wave-uniform: the common window costs ~10 vector instructions)
                    GRK_K3_STORE(reinterpret_cast<u32_any*>(out + j + lane4), win);
                    s += 2048u; j += 256u;
                    continue;
                }
                if (nb == 4u) GRK_K3_STORE(reinterpret_cast<u32_any*>(out + j + lane4), win);
                else {
#pragma unroll 1
                    for (uint32_t k = 0; k < nb; ++k) out[j + lane4 + k] = (uint8_t)(win >> (8u * k));
                }
                s += 8u * avail; j += avail;
                continue;
            }
            const uint32_t F = (uint32_t)__ffsll((long long)ballot) - 1u;
            const uint32_t zF = (uint32_t)__builtin_amdgcn_readlane((int)z, (int)F);
            const uint32_t winF = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)F);
            const uint32_t b = (uint32_t)(__ffs((int)zF) - 1) >> 3;
            if ((uint32_t)lane < F) GRK_K3_STORE(reinterpret_cast<u32_any*>(out + j + lane4), win);
            const uint32_t p7 = s + 32u * F + 8u * (b + 1u);         // where the 7-bit byte after the 0xFF starts
            // the raw bits from p7 on -- the rest of lane F's window and the next lane's (32 bits at least: the 7-bit byte and the look
            // at the byte behind it) -- are in registers already; r01-r05 went back to LDS for them twice per event
            uint32_t winN;
            if (F < 63u) winN = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(F + 1u));
            else {
                const uint32_t sn = s + 2048u;
                winN = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_alignbit(ms_raw[(sn >> 5) + 1u], ms_raw[sn >> 5], sn));
            }
            const uint64_t tail = (((uint64_t)winN << 32) | winF) >> (8u * (b + 1u));
            j += 4u * F;
            if (p7 == ms_bits) {                                     // the stream ends with the 0xFF: dropped (ms_terminate)
                if ((uint32_t)lane < b) out[j + lane] = (uint8_t)(winF >> (8u * lane));
                ms_len = j + b; done = true;
                break;
            }
            if ((uint32_t)lane <= b) out[j + lane] = (uint8_t)(winF >> (8u * lane));
            j += b + 1u;
            if (p7 + 7u > ms_bits) {                                 // incomplete 7-bit byte: padded with ones, never 0xFF
                const uint32_t rem = ms_bits - p7;
                const uint32_t fin = ((uint32_t)tail & ((1u << rem) - 1u)) | ((((1u << (7u - rem)) - 1u) << rem) & 0x7Fu);
                if (lane == 0) out[j] = (uint8_t)fin;
                ms_len = j + 1u; done = true;
                break;
            }
            const uint32_t b7 = (uint32_t)tail & 0x7Fu;
            if (lane == 0) out[j] = (uint8_t)b7;
            j += 1u;
            s = p7 + 7u;
            // A run of 0xFF bytes (flat areas whose MagSgn values are all ones: pixel value 0 is -128 after the DC shift -- the LL
            // band of a black frame is nothing else): every one of them is an event, and an event through the window above costs
            // a 256-byte look.  While the NEXT byte is 0xFF again and its 7-bit follower is whole, take the pair with one 15-bit
            // look (r04: K3 of an all-zero 8K frame 0.59 ms, all of it this loop in 48 LL blocks; profiles/r04_small_frames.txt);
            // everything else -- the stream's end, a partial follower -- goes back through the general path.
            // (one look decides whether the run loop is entered at all -- r05: with the loop directly behind the event K3 of dense
            //  content, which never takes it, was 0.7 % slower, and 2.2 % slower than without the loop: code placement, the registers
            //  and spills are the same; profiles/r05_k3_pairs.txt)
            if (s + 15u <= ms_bits && ((uint32_t)(tail >> 7) & 0xFFu) == 0xFFu)
            while (s + 15u <= ms_bits) {
                const uint32_t two = (uint32_t)__builtin_amdgcn_readfirstlane((int)get_bits(ms_raw, s, 15u));
                if ((two & 0xFFu) != 0xFFu) break;
                if (lane == 0) { out[j] = 0xFFu; out[j + 1u] = (uint8_t)(two >> 8); }
                j += 2u; s += 15u;
            }
        }
        if (!done) {
            const uint32_t rem = ms_bits - s;
            if (rem) {
                const uint32_t fin = get_bits(ms_raw, s, rem) | ((((1u << (8u - rem)) - 1u) << rem) & 0xFFu);
                if (fin != 0xFFu) {
                    if (lane == 0) out[j] = (uint8_t)fin;
                    j += 1u;
                }
            }
            ms_len = j;
        }
    }
    const uint32_t total = ms_len + mel_len + vcount + 1;
    if (lane == 0) { a.lengths[gid] = total; a.offsets[gid] = base_off; }
    if (total > len_ub) {                                            // (cannot happen: the bound counts every stuffing bit)
        if (lane == 0) atomicOr(reinterpret_cast<unsigned int*>(a.alloc), 1u);
        return;
    }
    __syncthreads();                                                 // lane 0's last MEL byte
#pragma unroll 1
    for (uint32_t i = lane; i < mel_len; i += 64) out[ms_len + i] = mel_buf[i < 250 ? i : 249];

    // ---- B4: VLC bytes 0 .. nv - 1, byte i at out[total - 2 - i] (emit_vlc of the model); the first one carries Scup's low nibble
    {                                                                // the staged bytes, four per lane, reversed
        uint8_t* const last = out + total - 2;
#pragma unroll 1
        for (uint32_t k = 4u * (uint32_t)lane; k < nv; k += 256u) {
            uint32_t word = *reinterpret_cast<const uint32_t*>(vst + k);
            if (k == 0) word = (word & ~0xFu) | (scup & 0xFu);
            const uint32_t n = min(nv - k, 4u);
            if (n == 4u) GRK_K3_STORE(reinterpret_cast<u32_any*>(last - 3 - (int)k), __builtin_bswap32(word));
            else {
#pragma unroll 1
                for (uint32_t t = 0; t < n; ++t) *(last - (int)(k + t)) = (uint8_t)(word >> (8u * t));
            }
        }
    }
    if (lane == 0) {
        if (vextra) out[total - 2 - nv] = (uint8_t)(nv == 0 ? ((vacc & 0xF0) | (scup & 0xF)) : vacc);
        out[total - 1] = (uint8_t)(scup >> 4);
    }
}

// ROOM: the launch runs beside the NEXT frame's DWT level 0 (the K3 launches of a pipelined encode, on the low-priority side streams).
// A level-0 workgroup is four waves of 72 registers and 24 KB of LDS; K3 at five waves per SIMD leaves a CU neither the registers
// (r05 first half: 96 x 5, 32 free) nor the LDS (7 KB per wave x 20, 17 KB free), so a level-0 workgroup -- the critical path of the
// pipeline -- finds room on a CU only when waves of K3 have retired on EVERY SIMD of it.  The ROOM instance claims 104 registers (it
// uses 83): four waves per SIMD, 96 registers and 48 KB of LDS free for the DWT.  Alone it is 10 % slower (0.295 against 0.267 ms for
// the 8K frame), beside the DWT the pipelined step goes from 0.412 to 0.367-0.376 ms per frame on the same box.  What else was
// measured (profiles/r05_k3_pairs.txt 6-7, 10-11): a claim of 112 (four waves, 64 registers free: no room for level 0) 0.423, of 128
// 0.438, of 136 (three waves, 104 free) 0.405; FIVE waves with guaranteed room -- this kernel at 83 registers, LDS capacities cut to
// 6.4 KB per wave, level 0 built at 60 registers -- 0.395-0.405: room is necessary, and four waves of K3 beside it is the balance;
// capping K3's waves with LDS instead takes the LDS the DWT needs (0.449), CU masks lose outright (profiles/r05_sq_overlapped_8k.txt).
template <bool IRREV, bool H16, bool ROOM>
__global__ __launch_bounds__(64) void ht_encode_kernel(HtArgs a, HtLds L, uint32_t class_id)
{
    if constexpr (ROOM) asm volatile("" ::: "v103");
    ht_encode_block<IRREV, H16>(a, blockIdx.x % a.sel_count, blockIdx.x / a.sel_count, L, class_id);
}


// The blocks the launch above handed over (raw streams outgrew their LDS): worst-case buffers, a fixed grid that walks
// the list.  With nothing on the list -- the normal case -- every workgroup leaves at once.
template <bool IRREV, bool H16>
__global__ __launch_bounds__(64) void ht_encode_fallback_kernel(HtArgs a, HtLds L, uint32_t class_id)
{
    const uint32_t count = (uint32_t)a.alloc[2 + class_id];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t id = a.ovf_list[a.ovf_base + i];
        ht_encode_block<IRREV, H16>(a, id % a.sel_count, id / a.sel_count, L, class_id);
        __syncthreads();
    }
}

} // namespace

static std::atomic<bool> g_tables_ready[16];                 // (zero-initialised: false)
static std::mutex g_tables_mu;                               // first use from several host threads at once (node workers on one device)
static const uint32_t* g_vlc_tab_dev[16] = {nullptr};     // device address of g_vlc_enc per device (a kernel argument: a scalar base register)

static hipError_t upload_tables()
{
    // first-row table as generated; other rows re-indexed by the neighbour flags the kernel computes
    static uint32_t enc[4096];
    auto entry = [](uint32_t t) {          // generated (cwd << 8 | len << 4 | e_k) -> the kernel's layout
        const uint32_t ek = t & 15u;
        return ((t >> 8) << 25) | (((t >> 4) & 7u) << 4) | (ek & 1u) | (((ek >> 1) & 1u) << 8) | (((ek >> 2) & 1u) << 16) | ((ek >> 3) << 24);
    };
    for (uint32_t i = 0; i < 2048; ++i) {
        enc[i] = entry(HT_VLC_ENC0[i]);
        const uint32_t ctx = i >> 8, n_w = ctx & 1u, rl = (ctx >> 1) & 1u, n_e = (ctx >> 2) & 1u;
        const uint32_t c_q = (n_w ? 0u : 1u) | (rl << 1) | ((n_e ? 0u : 1u) << 2);
        enc[2048 + i] = entry(HT_VLC_ENC1[(c_q << 8) | (i & 0xFFu)]);
    }
    static uint2 uv[64];
    for (uint32_t u = 0; u < 64; ++u) {
        uint32_t pre = 0, pl = 0, suf = 0, sl = 0;
        if (u < 32) {                                   // ojph_block_encoder.cpp:189-210
            if (u <= 2)      { pre = u; pl = u; }
            else if (u <= 4) { pre = 4; pl = 3; suf = u - 3; sl = 1; }
            else             { pre = 0; pl = 3; suf = u - 5; sl = 5; }
        } else if (u == 33 || u == 34) { pre = u - 33; pl = 1; }
        uv[u].x = (pre << 8) | (suf << 16);
        uv[u].y = (pl << 8) | (sl << 16);
    }
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlc_enc), enc, sizeof(enc), 0, hipMemcpyHostToDevice);
    if (e != hipSuccess) return e;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_uvlc), uv, sizeof(uv), 0, hipMemcpyHostToDevice);
}

// LDS words of a launch whose largest block has `samples` samples in `quads` quads and exponent kmax
// capped = false: the worst case (m_n <= U_q <= Kmax + 2 inside the contract; cwd <= 7, UVLC prefix <= 3, suffix <= 5
// bits per quad).  capped = true: what real content needs with room to spare -- reversible: 8 bits per sample on
// average for 8-bit content (Kmax <= 11), Kmax - 3 beyond; quantised (irreversible) coefficients: 8 bits whatever the
// exponent (the default step sizes leave ~3 bits per sample of a 16-bit image); 10 VLC bits per quad.
static void ht_lds_layout(uint32_t samples, uint32_t quads, uint32_t kmax, bool capped, bool irrev, HtLds& L, size_t& bytes)
{
    const uint32_t per_sample = !capped ? kmax + 2u : irrev ? std::min(kmax + 2u, 8u) : std::min(kmax + 2u, kmax <= 11u ? 8u : kmax - 3u);
    const uint32_t ms_bits = samples * per_sample;
    const uint32_t vlc_bits = quads * (capped ? 10u : 15u) + 4u;
    L.ms_cap_bits = ms_bits;
    L.vlc_cap_bits = vlc_bits;
    L.ms_words = ((ms_bits + 31u) / 32u + 4u + 3u) & ~3u;           // slack: or_bits64 / window reads touch two words beyond;
    L.vlc_words = ((vlc_bits + 31u) / 32u + 4u + 3u) & ~3u;         // multiples of 4 words: cleared as uint4
    // behind the raw streams: the UVLC table (64 x 8 bytes) while phase A runs, then the staged VLC bytes (phase B1: the stuffed
    // bytes of the stream's capacity) in the same place; then 256 MEL bytes.  (A raw stream's windows read up to 65 words past its
    // end: the two areas are at least 192 words.)
    L.stage_bytes = std::max<uint32_t>(((vlc_bits / 7u + 16u) + 15u) & ~15u, 512u);
    bytes = (size_t)(L.ms_words + L.vlc_words) * 4u + L.stage_bytes + 256u;
}

size_t ht_lds_bytes(uint32_t samples, uint32_t quads, uint32_t kmax)
{
    HtLds L{}; size_t n;
    ht_lds_layout(samples, quads, kmax, false, false, L, n);
    return n;
}

static hipError_t ensure_tables(const uint32_t** tab = nullptr)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!g_tables_ready[dev].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_tables_mu);
        if (!g_tables_ready[dev].load(std::memory_order_relaxed)) {
            e = upload_tables();
            if (e != hipSuccess) return e;
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_vlc_enc));
            if (e != hipSuccess) return e;
            g_vlc_tab_dev[dev] = static_cast<const uint32_t*>(p);
            g_tables_ready[dev].store(true, std::memory_order_release);
        }
    }
    if (tab) *tab = g_vlc_tab_dev[dev];
    return hipSuccess;
}

hipError_t launch_ht_alloc_init(const HtArgs& a, hipStream_t s)
{
    hipError_t e = ensure_tables();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ht_alloc_init_kernel, dim3(1), dim3(64), 0, s, a.alloc, a.chunk_units);
    return hipGetLastError();
}

hipError_t launch_ht_classes(const HtArgs& a, uint32_t first, uint32_t last, hipStream_t s)
{
    const uint32_t* vlc_tab = nullptr;
    hipError_t e = ensure_tables(&vlc_tab);
    if (e != hipSuccess) return e;
    // one launch per block class (HtClass): the dynamic LDS size is what fixes the occupancy, and the
    // few high-Kmax blocks of the low resolutions would otherwise cost every block a wave per SIMD
    for (uint32_t k = first; k < last && k < a.num_classes; ++k) {
        const HtClass& c = a.classes[k];
        if (c.count == 0) continue;
        // capped LDS when that buys occupancy (waves per CU = 160 KiB / LDS per wave, at most 32), else worst-case buffers
        HtLds full{}, cap{};
        size_t shmem_full, shmem_cap;
        ht_lds_layout(c.max_samples, c.max_quads, c.max_kmax, false, false, full, shmem_full);
        ht_lds_layout(c.max_samples, c.max_quads, c.cap_kmax, true, a.irreversible != 0, cap, shmem_cap);
        auto waves = [](size_t lds) { return std::min<size_t>(32, (160u << 10) / std::max<size_t>(lds, 1)); };
        const bool use_cap = a.ovf_list && waves(shmem_cap) > waves(shmem_full);
        const HtLds& L = use_cap ? cap : full;
        const size_t shmem = use_cap ? shmem_cap : shmem_full;
        HtArgs b = a;
        b.sel = c.sel; b.sel_count = c.count;
        b.ovf_base = c.ovf_base;
        b.vlc_tab = vlc_tab;
        const uint32_t grid = c.count * a.ntiles;
#define GRK_HT(KERNEL, G, SH, LL)                                                                                                   \
        do {                                                                                                                        \
            if (a.h16 && !a.irreversible) hipLaunchKernelGGL((KERNEL<false, true>), dim3(G), dim3(64), SH, s, b, LL, k);         \
            else if (a.irreversible)      hipLaunchKernelGGL((KERNEL<true, false>), dim3(G), dim3(64), SH, s, b, LL, k);          \
            else                          hipLaunchKernelGGL((KERNEL<false, false>), dim3(G), dim3(64), SH, s, b, LL, k);         \
        } while (0)
        // (the packed 8-bit kernel first: its place in the code object does not move when the others change)
        if (a.h16 && !a.irreversible && a.room) hipLaunchKernelGGL((ht_encode_kernel<false, true, true>), dim3(grid), dim3(64), shmem, s, b, L, k);
        else if (a.h16 && !a.irreversible)      hipLaunchKernelGGL((ht_encode_kernel<false, true, false>), dim3(grid), dim3(64), shmem, s, b, L, k);
        // (r06: a ROOM instance of the 32-bit kernels was measured beside the level-0 DWT that now fits the registers it would leave
        //  -- kernels_dwt.hip GEN = false, 83 / 84 registers --: cfg3's step 0.640 / 0.647 without, 0.642 / 0.645 with: not kept)
        else if (a.irreversible)                hipLaunchKernelGGL((ht_encode_kernel<true, false, false>), dim3(grid), dim3(64), shmem, s, b, L, k);
        else                                    hipLaunchKernelGGL((ht_encode_kernel<false, false, false>), dim3(grid), dim3(64), shmem, s, b, L, k);
        // the blocks that outgrew the capped buffers: a small fixed grid walks the list (with nothing on it -- natural
        // images at the top resolution -- its workgroups leave at once)
        if (use_cap) GRK_HT(ht_encode_fallback_kernel, std::min<uint32_t>(grid, 1024u), shmem_full, full);
#undef GRK_HT
    }
    return hipGetLastError();
}

hipError_t launch_ht_encode(const HtArgs& a, hipStream_t s)
{
    hipError_t e = launch_ht_alloc_init(a, s);
    if (e != hipSuccess) return e;
    return launch_ht_classes(a, 0, a.num_classes, s);
}

} // namespace grk_amd
