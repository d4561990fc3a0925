// grok_amd/csrc/pk16.h -- two int16 in one register (v_pk_add / sub / ashr: one instruction, two samples), gfx950.
// The 5/3 lifting steps are sums, differences and floor shifts, so while nothing leaves 16 bits the halves are exactly what the
// 32-bit form computes.  Who vouches for the range: the host for the forward transform (context.hip pk16_level_ok: 8-bit
// pixels bound every level), the producers' range flags for the inverse (kernels_htdec.hip / kernels_idwt.hip: every
// coefficient and every intermediate LL inside +-2047, or status bit 3 and the decode is done again in 32 bits).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace grk_amd {

typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ uint32_t as_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));      // what the 8-byte buffer loads / stores carry

// v_perm_b32 selectors over (hi = S0 : lo = S1): the low halves / the high halves of two registers side by side
constexpr uint32_t kSelLoLo = 0x05040100u;      // (S1.lo | S0.lo << 16)
constexpr uint32_t kSelHiHi = 0x07060302u;      // (S1.hi | S0.hi << 16)

// a buffer descriptor over everything from p on (offsets are 32-bit: a plane is at most 2^31 samples), or of zero length --
// what goes through that one the memory pipeline drops
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_from(const void* p, bool none = false)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, none ? 0 : -1, 0x00020000);
}

// largest magnitude the packed inverse transform takes in (coefficients and intermediate LL): with it no sum of the
// horizontal + vertical synthesis, of the inverse RCT and of the DC shift leaves 16 bits (7.5 M + 4, 15.6 M + 128 < 32768)
constexpr int32_t kPkDecodeBound = 2047;

} // namespace grk_amd
