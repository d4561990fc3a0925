/* oracle/ht_wave_model.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU model of "phase B" of the wave-per-code-block HT encoder kernel
 * (grok_amd/csrc/kernels_ht.hip): given the RAW (un-stuffed) MagSgn and VLC bit streams of a
 * block, produce the final byte string with all 64 "lanes" working on independent pieces.
 * It exists so that the parallel formulation can be proven equal to the serial writers of the
 * oracle (orc_ht_encode_sm, itself pinned to ojph_encode_codeblock) on a machine without a GPU;
 * the kernel is a transliteration of this file (same variable names).
 *
 * Idea: byte-stuffing only changes where byte boundaries fall after "events":
 *   MagSgn: a byte equal to 0xFF makes the NEXT byte 7 bits wide;
 *   VLC   : a byte that follows a byte > 0x8F and whose first 7 bits are all ones IS 7 bits wide.
 * Events can only occur where the raw stream has 8 (7) consecutive one bits, which is rare, so
 *   1. a wave-uniform "walker" hops from event to event: 64 lanes test 64 raw words (2048 bits)
 *      per round for candidate positions of the current byte phase (ballot + ctz);
 *      every event sets one bit in a bitmap over OUTPUT byte indices ("this byte has 7 bits");
 *   2. a prefix popcount of that bitmap tells each output byte how many stuffing bits precede
 *      it, i.e. where its bits start in the raw stream: start(j) = 8 j - #marks below j;
 *   3. all lanes then emit output bytes independently.
 */
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define LANES 64

static uint32_t rd(const uint32_t* raw, uint32_t nwords, int64_t i, uint32_t before)
{
    if (i < 0) return before;
    return (uint64_t)i < nwords ? raw[i] : 0u;
}
static uint32_t get_bits(const uint32_t* raw, uint32_t nwords, uint32_t pos, uint32_t n)
{
    uint64_t v = rd(raw, nwords, pos >> 5, 0) | ((uint64_t)rd(raw, nwords, (pos >> 5) + 1, 0) << 32);
    return (uint32_t)(v >> (pos & 31)) & ((1u << n) - 1);
}
static uint32_t popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }

/* walker shared by both streams. vlc_rule = 0: MagSgn (candidate = 8 ones, mark byte j+1);
 * 1: VLC (candidate = 7 ones after a byte > 0x8F, mark byte j). Returns number of events. */
static uint32_t walk(const uint32_t* raw, uint32_t nwords, uint32_t nbits, int vlc_rule,
                     uint32_t* marks, uint32_t* last_p)
{
    uint32_t K = 0, s = 0;
    const uint32_t need = vlc_rule ? 7 : 8;
    *last_p = 0;
    while (s + need <= nbits) {
        const uint32_t B = s >> 5, phase = s & 7;
        uint32_t hit[LANES];
        uint64_t ballot = 0;
        for (int l = 0; l < LANES; ++l) {                      /* ---- one lane each */
            const int64_t i = (int64_t)B + l;
            const uint64_t hi = rd(raw, nwords, i, 0) | ((uint64_t)rd(raw, nwords, i + 1, 0) << 32);
            uint32_t cand;
            if (!vlc_rule) {
                uint64_t c = hi & (hi >> 1); c &= c >> 2; c &= c >> 4;
                cand = (uint32_t)c;
            } else {
                uint64_t c = hi & (hi >> 1); c &= c >> 2; c &= c >> 3;
                const uint64_t lo = rd(raw, nwords, i - 1, 0xFFFFFFFFu) | ((uint64_t)rd(raw, nwords, i, 0) << 32);
                const uint64_t pv = (lo >> 31) & ((lo >> 30) | (lo >> 29) | (lo >> 28));
                cand = (uint32_t)c & (uint32_t)pv;
            }
            uint32_t m = 0x01010101u << phase;
            if (l == 0) m &= 0xFFFFFFFFu << (s & 31);
            hit[l] = cand & m;
            if (hit[l]) ballot |= 1ull << l;
        }
        if (!ballot) { s = 32 * (B + LANES) + phase; continue; }
        const int L = __builtin_ctzll(ballot);
        const uint32_t p = 32 * (B + (uint32_t)L) + (uint32_t)__builtin_ctz(hit[L]);
        const uint32_t j = (p + K) >> 3;                         /* output index of the byte at p */
        const uint32_t mj = vlc_rule ? j : j + 1;
        if (marks) marks[mj >> 5] |= 1u << (mj & 31);
        ++K; *last_p = p;
        s = p + 15;
    }
    return K;
}

/* returns total length; out must hold it */
int32_t orc_ht_model_phase_b(const uint32_t* ms_raw, uint32_t ms_bits, const uint32_t* vlc_raw, uint32_t vlc_bits,
                             const uint8_t* mel_bytes, const int* mel_state, uint8_t* out)
{
    const uint32_t msw = (ms_bits + 31) / 32, vw = (vlc_bits + 31) / 32;
    const uint32_t MW = (ms_bits / 7 + 64) / 32 + 2, VW = (vlc_bits / 7 + 64) / 32 + 2;   /* output bytes <= bits/7 */
    uint32_t* marks = (uint32_t*)calloc(MW, 4);  uint32_t* pref = (uint32_t*)calloc(MW + 1, 4);
    uint32_t* vmarks = (uint32_t*)calloc(VW, 4); uint32_t* vpref = (uint32_t*)calloc(VW + 1, 4);

    /* ---- B1: MagSgn events + termination (ojph ms_encode / ms_terminate) */
    uint32_t last_p;
    const uint32_t K = walk(ms_raw, msw, ms_bits, 0, marks, &last_p);
    uint32_t pos, limit, nfull;
    if (K && last_p + 15 > ms_bits) { pos = last_p + 8; limit = 7; nfull = (pos + K - 1) >> 3; }
    else { const uint32_t s0 = K ? last_p + 15 : 0; pos = s0 + 8 * ((ms_bits - s0) / 8); limit = 8; nfull = (pos + K) >> 3; }
    const uint32_t rem = ms_bits - pos;
    uint32_t ms_len, final_byte = 0, has_final = 0;
    if (rem > 0) {
        final_byte = get_bits(ms_raw, msw, pos, rem) | ((((1u << (limit - rem)) - 1u) << rem) & 0xFF);
        has_final = final_byte != 0xFF;
        ms_len = nfull + has_final;
    } else {
        ms_len = (limit == 7) ? nfull - 1 : nfull;
    }
    const uint32_t ms_emit = nfull < ms_len ? nfull : ms_len;

    /* ---- B2: VLC events + tail */
    uint32_t vlast;
    const uint32_t Kv = walk(vlc_raw, vw, vlc_bits, 1, vmarks, &vlast);
    const uint32_t vs0 = Kv ? vlast + 7 : 0;
    const uint32_t vposr = vs0 + 8 * ((vlc_bits - vs0) / 8);
    const uint32_t vused = vlc_bits - vposr;
    const uint32_t vacc = vused ? get_bits(vlc_raw, vw, vposr, vused) : 0;
    const uint32_t nv = (vposr + Kv) >> 3;                       /* complete VLC data bytes */

    /* ---- B3: MEL/VLC termination (terminate_mel_vlc) -- one lane */
    uint32_t mel_pos = (uint32_t)mel_state[0];
    int mel_acc = mel_state[1], mel_left = mel_state[2], mel_run = mel_state[3];
    uint8_t mel_tail[2]; uint32_t mel_tail_n = 0;
    if (mel_run > 0) {
        mel_acc = (mel_acc << 1) | 1;
        if (--mel_left == 0) { mel_tail[mel_tail_n++] = (uint8_t)mel_acc; mel_left = (mel_acc == 0xFF) ? 7 : 8; mel_acc = 0; }
    }
    uint32_t vextra = 0, vextra_byte = 0;
    {
        const int macc = mel_acc << mel_left;
        const int mel_mask = (0xFF << mel_left) & 0xFF;
        const int vlc_mask = 0xFF >> (8 - (int)vused);
        if ((mel_mask | vlc_mask) != 0) {
            const int fuse = macc | (int)vacc;
            if ((((fuse ^ macc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && nv >= 1)
                mel_tail[mel_tail_n++] = (uint8_t)fuse;
            else { mel_tail[mel_tail_n++] = (uint8_t)macc; vextra = 1; vextra_byte = vacc; }
        }
    }
    const uint32_t mel_len = mel_pos + mel_tail_n;
    const uint32_t vcount = nv + vextra;
    const uint32_t total = ms_len + mel_len + vcount + 1;

    /* ---- B4: prefix popcounts of the mark bitmaps */
    for (uint32_t i = 0; i < MW; ++i) pref[i + 1] = pref[i] + popc(marks[i]);
    for (uint32_t i = 0; i < VW; ++i) vpref[i + 1] = vpref[i] + popc(vmarks[i]);

    /* ---- B5: emission, every output byte independent (lanes stride over dwords on the GPU) */
    for (uint32_t j = 0; j < ms_emit; ++j) {
        const uint32_t k = pref[j >> 5] + popc(marks[j >> 5] & ((1u << (j & 31)) - 1u));
        const uint32_t is7 = (marks[j >> 5] >> (j & 31)) & 1;
        out[j] = (uint8_t)get_bits(ms_raw, msw, 8 * j - k, is7 ? 7 : 8);
    }
    if (has_final) out[ms_len - 1] = (uint8_t)final_byte;
    memcpy(out + ms_len, mel_bytes, mel_pos);
    for (uint32_t i = 0; i < mel_tail_n; ++i) out[ms_len + mel_pos + i] = mel_tail[i];
    for (uint32_t j = 0; j < nv; ++j) {
        const uint32_t k = vpref[j >> 5] + popc(vmarks[j >> 5] & ((1u << (j & 31)) - 1u));
        const uint32_t is7 = (vmarks[j >> 5] >> (j & 31)) & 1;
        out[total - 2 - j] = (uint8_t)get_bits(vlc_raw, vw, 8 * j - k, is7 ? 7 : 8);
    }
    if (vextra) out[total - 2 - nv] = (uint8_t)vextra_byte;
    const uint32_t scup = mel_len + vcount + 1;
    out[total - 1] = (uint8_t)(scup >> 4);
    out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
    free(marks); free(pref); free(vmarks); free(vpref);
    return (int32_t)total;
}

/* ---- r03: phase B without bitmaps ("speculative windows", kernels_ht.hip emit_ms / emit_vlc) -------------------------------
 * The walker above finds the events first and the emission looks every byte's start up afterwards.  The second form does
 * both at once: a window of 64 lanes x 4 output bytes is cut out of the raw stream as if no event fell into it; every lane
 * tests its own four bytes; with no event in the window (ballot == 0) all 256 bytes are final, otherwise everything before
 * the first event is, the event's 7-bit byte is handled on the spot and the next window starts right behind it.
 * MagSgn needs no other pass at all (its byte count falls out at the end); VLC bytes are stored backwards from the END of
 * the block, so their count has to be known first: the walker above still counts them (marks == NULL). */
static uint32_t get32(const uint32_t* raw, uint32_t nwords, uint32_t pos)
{
    uint64_t v = rd(raw, nwords, pos >> 5, 0) | ((uint64_t)rd(raw, nwords, (pos >> 5) + 1, 0) << 32);
    return (uint32_t)(v >> (pos & 31));
}
static uint32_t valid_mask(uint32_t nb, uint32_t flags) { return nb >= 4 ? flags : ((1u << (8 * nb)) - 1u) & flags; }

static uint32_t emit_ms(const uint32_t* raw, uint32_t nwords, uint32_t nbits, uint8_t* out)
{
    uint32_t s = 0, j = 0;
    while (s + 8 <= nbits) {
        const uint32_t avail = (nbits - s) >> 3;                 /* whole bytes left if no event comes */
        uint32_t win[LANES], z[LANES], nbl[LANES];
        uint64_t ballot = 0;
        for (int l = 0; l < LANES; ++l) {                        /* ---- one lane each */
            win[l] = get32(raw, nwords, s + 32u * l);
            nbl[l] = avail > 4u * l ? (avail - 4u * l < 4 ? avail - 4u * l : 4) : 0;
            const uint32_t x = ~win[l];
            z[l] = (x - 0x01010101u) & win[l] & valid_mask(nbl[l], 0x80808080u);      /* lowest flag = first 0xFF byte (exact) */
            if (z[l]) ballot |= 1ull << l;
        }
        if (!ballot) {
            for (int l = 0; l < LANES; ++l)
                for (uint32_t b = 0; b < nbl[l]; ++b) out[j + 4u * l + b] = (uint8_t)(win[l] >> (8 * b));
            const uint32_t n = avail < 256 ? avail : 256;
            s += 8 * n; j += n;
            continue;
        }
        const uint32_t F = (uint32_t)__builtin_ctzll(ballot), b = (uint32_t)__builtin_ctz(z[F]) >> 3;
        for (uint32_t l = 0; l < F; ++l)
            for (uint32_t k = 0; k < 4; ++k) out[j + 4u * l + k] = (uint8_t)(win[l] >> (8 * k));
        const uint32_t p7 = s + 32u * F + 8u * (b + 1);          /* where the 7-bit byte after the 0xFF starts */
        if (p7 == nbits) {                                       /* the stream ends with the 0xFF: dropped (fb_finish) */
            for (uint32_t k = 0; k < b; ++k) out[j + 4u * F + k] = (uint8_t)(win[F] >> (8 * k));
            return j + 4u * F + b;
        }
        for (uint32_t k = 0; k <= b; ++k) out[j + 4u * F + k] = (uint8_t)(win[F] >> (8 * k));
        j += 4u * F + b + 1;
        if (p7 + 7 > nbits) {                                    /* incomplete 7-bit byte: padded with ones, never 0xFF */
            const uint32_t rem = nbits - p7;
            out[j++] = (uint8_t)(get_bits(raw, nwords, p7, rem) | ((((1u << (7 - rem)) - 1u) << rem) & 0x7F));
            return j;
        }
        out[j++] = (uint8_t)get_bits(raw, nwords, p7, 7);
        s = p7 + 7;
    }
    const uint32_t rem = nbits - s;
    if (rem) {
        const uint32_t fin = get_bits(raw, nwords, s, rem) | ((((1u << (8 - rem)) - 1u) << rem) & 0xFF);
        if (fin != 0xFF) out[j++] = (uint8_t)fin;
    }
    return j;
}

/* VLC bytes 0 .. nv-1, byte i stored at last[-i] */
static void emit_vlc(const uint32_t* raw, uint32_t nwords, uint32_t nv, uint8_t* last)
{
    uint32_t s = 0, i = 0, prev = 0xFF;
    while (i < nv) {
        const uint32_t left = nv - i;
        uint32_t win[LANES], z[LANES], nbl[LANES];
        uint64_t ballot = 0;
        for (int l = 0; l < LANES; ++l) {
            win[l] = get32(raw, nwords, s + 32u * l);
            nbl[l] = left > 4u * l ? (left - 4u * l < 4 ? left - 4u * l : 4) : 0;
        }
        for (int l = 0; l < LANES; ++l) {
            const uint32_t prevsrc = l ? win[l - 1] : prev << 24;
            const uint32_t pw = (win[l] << 8) | (prevsrc >> 24);       /* byte k of pw = the byte before byte k of win */
            const uint32_t e = (win[l] & 0x7F7F7F7Fu) + 0x01010101u;   /* bit 7 of a byte: its low 7 bits are ones */
            /* flag at bit 4 of a byte (right shifts are the cheap ones on the GPU): bit 7 of the byte before and any of its
               bits 6..4 (it is > 0x8F), and the low 7 bits of this one are ones */
            z[l] = (pw >> 3) & ((pw >> 2) | (pw >> 1) | pw) & (e >> 3) & valid_mask(nbl[l], 0x10101010u);
            if (z[l]) ballot |= 1ull << l;
        }
        if (!ballot) {
            for (int l = 0; l < LANES; ++l)
                for (uint32_t b = 0; b < nbl[l]; ++b) last[-(int32_t)(i + 4u * l + b)] = (uint8_t)(win[l] >> (8 * b));
            const uint32_t n = left < 256 ? left : 256;
            prev = win[LANES - 1] >> 24;
            s += 8 * n; i += n;
            continue;
        }
        const uint32_t F = (uint32_t)__builtin_ctzll(ballot), b = (uint32_t)__builtin_ctz(z[F]) >> 3;
        for (uint32_t l = 0; l < F; ++l)
            for (uint32_t k = 0; k < 4; ++k) last[-(int32_t)(i + 4u * l + k)] = (uint8_t)(win[l] >> (8 * k));
        for (uint32_t k = 0; k < b; ++k) last[-(int32_t)(i + 4u * F + k)] = (uint8_t)(win[F] >> (8 * k));
        last[-(int32_t)(i + 4u * F + b)] = 0x7F;
        s += 32u * F + 8u * b + 7u; i += 4u * F + b + 1; prev = 0x7F;
    }
}

int32_t orc_ht_model_phase_b2(const uint32_t* ms_raw, uint32_t ms_bits, const uint32_t* vlc_raw, uint32_t vlc_bits,
                              const uint8_t* mel_bytes, const int* mel_state, uint8_t* out)
{
    const uint32_t msw = (ms_bits + 31) / 32, vw = (vlc_bits + 31) / 32;
    const uint32_t ms_len = emit_ms(ms_raw, msw, ms_bits, out);

    uint32_t vlast;
    const uint32_t Kv = walk(vlc_raw, vw, vlc_bits, 1, NULL, &vlast);
    const uint32_t vs0 = Kv ? vlast + 7 : 0;
    const uint32_t vposr = vs0 + 8 * ((vlc_bits - vs0) / 8);
    const uint32_t vused = vlc_bits - vposr;
    const uint32_t vacc = vused ? get_bits(vlc_raw, vw, vposr, vused) : 0;
    const uint32_t nv = (vposr + Kv) >> 3;

    uint32_t mel_pos = (uint32_t)mel_state[0];
    int mel_acc = mel_state[1], mel_left = mel_state[2], mel_run = mel_state[3];
    uint8_t mel_tail[2]; uint32_t mel_tail_n = 0;
    if (mel_run > 0) {
        mel_acc = (mel_acc << 1) | 1;
        if (--mel_left == 0) { mel_tail[mel_tail_n++] = (uint8_t)mel_acc; mel_left = (mel_acc == 0xFF) ? 7 : 8; mel_acc = 0; }
    }
    uint32_t vextra = 0, vextra_byte = 0;
    {
        const int macc = mel_acc << mel_left;
        const int mel_mask = (0xFF << mel_left) & 0xFF;
        const int vlc_mask = 0xFF >> (8 - (int)vused);
        if ((mel_mask | vlc_mask) != 0) {
            const int fuse = macc | (int)vacc;
            if ((((fuse ^ macc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && nv >= 1)
                mel_tail[mel_tail_n++] = (uint8_t)fuse;
            else { mel_tail[mel_tail_n++] = (uint8_t)macc; vextra = 1; vextra_byte = vacc; }
        }
    }
    const uint32_t mel_len = mel_pos + mel_tail_n;
    const uint32_t vcount = nv + vextra;
    const uint32_t total = ms_len + mel_len + vcount + 1;
    memcpy(out + ms_len, mel_bytes, mel_pos);
    for (uint32_t i = 0; i < mel_tail_n; ++i) out[ms_len + mel_pos + i] = mel_tail[i];
    emit_vlc(vlc_raw, vw, nv, out + total - 2);
    if (vextra) out[total - 2 - nv] = (uint8_t)vextra_byte;
    const uint32_t scup = mel_len + vcount + 1;
    out[total - 1] = (uint8_t)(scup >> 4);
    out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
    return (int32_t)total;
}

/* ---- third form (r05, what kernels_ht.hip runs for packed 8-bit content) ----------------------------------------------------------
 * The VLC stream is looked at ONCE: the speculative windows run forwards BEFORE the termination and put byte i of the segment at
 * stage[i]; the number of whole bytes and the bits left over -- what the termination needs, and what the second form got from the
 * counting walker -- fall out of the same pass; the staged bytes are copied out reversed at the end.  A byte is whole when 8 bits
 * are left for it, or 7 if it is a 7-bit byte (it follows a byte > 0x8F and its bits are all ones): the tail check below. */
static uint32_t stage_vlc(const uint32_t* raw, uint32_t nwords, uint32_t nbits, uint8_t* stage, uint32_t* vposr)
{
    uint32_t s = 0, i = 0, prev = 0xFF;
    while (1) {
        const uint32_t avail = (nbits - s) >> 3;
        if (avail == 0) break;
        uint32_t win[LANES], z[LANES], nbl[LANES];
        uint64_t ballot = 0;
        for (int l = 0; l < LANES; ++l) {
            win[l] = get32(raw, nwords, s + 32u * l);
            nbl[l] = avail > 4u * l ? (avail - 4u * l < 4 ? avail - 4u * l : 4) : 0;
        }
        for (int l = 0; l < LANES; ++l) {
            const uint32_t prevsrc = l ? win[l - 1] : prev << 24;
            const uint32_t pw = (win[l] << 8) | (prevsrc >> 24);
            const uint32_t e = (win[l] & 0x7F7F7F7Fu) + 0x01010101u;
            z[l] = (pw >> 3) & ((pw >> 2) | (pw >> 1) | pw) & (e >> 3) & valid_mask(nbl[l], 0x10101010u);
            if (z[l]) ballot |= 1ull << l;
        }
        if (!ballot) {
            for (int l = 0; l < LANES; ++l)
                for (uint32_t b = 0; b < nbl[l]; ++b) stage[i + 4u * l + b] = (uint8_t)(win[l] >> (8 * b));
            const uint32_t n = avail < 256 ? avail : 256;
            prev = (win[(n - 1) >> 2] >> (8 * ((n - 1) & 3))) & 0xFF;          /* the last byte handed out */
            s += 8 * n; i += n;
            continue;
        }
        const uint32_t F = (uint32_t)__builtin_ctzll(ballot), b = (uint32_t)__builtin_ctz(z[F]) >> 3;
        for (uint32_t l = 0; l < F; ++l)
            for (uint32_t k = 0; k < 4; ++k) stage[i + 4u * l + k] = (uint8_t)(win[l] >> (8 * k));
        for (uint32_t k = 0; k < b; ++k) stage[i + 4u * F + k] = (uint8_t)(win[F] >> (8 * k));
        stage[i + 4u * F + b] = 0x7F;
        s += 32u * F + 8u * b + 7u; i += 4u * F + b + 1; prev = 0x7F;
    }
    if (nbits - s == 7 && prev > 0x8F && get_bits(raw, nwords, s, 7) == 0x7F) { stage[i++] = 0x7F; s += 7; }
    *vposr = s;
    return i;
}

int32_t orc_ht_model_phase_b3(const uint32_t* ms_raw, uint32_t ms_bits, const uint32_t* vlc_raw, uint32_t vlc_bits,
                              const uint8_t* mel_bytes, const int* mel_state, uint8_t* out)
{
    const uint32_t msw = (ms_bits + 31) / 32, vw = (vlc_bits + 31) / 32;
    static uint8_t stage[8192];
    uint32_t vposr;
    const uint32_t nv = stage_vlc(vlc_raw, vw, vlc_bits, stage, &vposr);
    const uint32_t vused = vlc_bits - vposr;
    const uint32_t vacc = vused ? get_bits(vlc_raw, vw, vposr, vused) : 0;

    uint32_t mel_pos = (uint32_t)mel_state[0];
    int mel_acc = mel_state[1], mel_left = mel_state[2], mel_run = mel_state[3];
    uint8_t mel_tail[2]; uint32_t mel_tail_n = 0;
    if (mel_run > 0) {
        mel_acc = (mel_acc << 1) | 1;
        if (--mel_left == 0) { mel_tail[mel_tail_n++] = (uint8_t)mel_acc; mel_left = (mel_acc == 0xFF) ? 7 : 8; mel_acc = 0; }
    }
    uint32_t vextra = 0, vextra_byte = 0;
    {
        const int macc = mel_acc << mel_left;
        const int mel_mask = (0xFF << mel_left) & 0xFF;
        const int vlc_mask = 0xFF >> (8 - (int)vused);
        if ((mel_mask | vlc_mask) != 0) {
            const int fuse = macc | (int)vacc;
            if ((((fuse ^ macc) & mel_mask) | ((fuse ^ (int)vacc) & vlc_mask)) == 0 && fuse != 0xFF && nv >= 1)
                mel_tail[mel_tail_n++] = (uint8_t)fuse;
            else { mel_tail[mel_tail_n++] = (uint8_t)macc; vextra = 1; vextra_byte = vacc; }
        }
    }
    const uint32_t ms_len = emit_ms(ms_raw, msw, ms_bits, out);
    const uint32_t mel_len = mel_pos + mel_tail_n;
    const uint32_t vcount = nv + vextra;
    const uint32_t total = ms_len + mel_len + vcount + 1;
    memcpy(out + ms_len, mel_bytes, mel_pos);
    for (uint32_t i = 0; i < mel_tail_n; ++i) out[ms_len + mel_pos + i] = mel_tail[i];
    for (uint32_t i = 0; i < nv; ++i) out[total - 2 - i] = stage[i];
    if (vextra) out[total - 2 - nv] = (uint8_t)vextra_byte;
    const uint32_t scup = mel_len + vcount + 1;
    out[total - 1] = (uint8_t)(scup >> 4);
    out[total - 2] = (uint8_t)((out[total - 2] & 0xF0) | (scup & 0xF));
    return (int32_t)total;
}
