// Shared device helpers for the gfx950 (CDNA4) kernels of the FastSpeech2 mel forward.
// Wave = 64 lanes; MFMA fragment maps follow /opt/skills/guides/cdna_hip_programming.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs2 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// bf16 is carried as raw 16-bit storage; arithmetic is always fp32.
struct bf16 {
    unsigned short v;
};

__host__ __device__ inline float bf16_to_f32(bf16 x) {
    union { uint32_t u; float f; } c;
    c.u = ((uint32_t)x.v) << 16;
    return c.f;
}
__host__ __device__ inline bf16 f32_to_bf16(float f) {  // round-to-nearest-even
    union { uint32_t u; float f; } c;
    c.f = f;
    uint32_t u = c.u;
    bf16 r;
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.v = (unsigned short)((u >> 16) | 0x40); return r; }  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    r.v = (unsigned short)(u >> 16);
    return r;
}

template <typename T> struct Num;
template <> struct Num<float> {
    __host__ __device__ static inline float to_f32(float x) { return x; }
    __host__ __device__ static inline float from_f32(float x) { return x; }
    static constexpr int kPer16B = 4;
};
template <> struct Num<bf16> {
    __host__ __device__ static inline float to_f32(bf16 x) { return bf16_to_f32(x); }
    __host__ __device__ static inline bf16 from_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
        bf16 r;
        const __bf16 h = (__bf16)x;  // v_cvt_pk_bf16_f32
        r.v = *(const unsigned short*)&h;
        return r;
#else
        return f32_to_bf16(x);
#endif
    }
    static constexpr int kPer16B = 8;
};

// 16-byte vector <-> fp32 lanes
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static inline void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
        f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {  // v_cvt_pk_bf16_f32 (RNE)
    const f32x2_t f = {lo, hi};
    const bf16x2_t v = __builtin_convertvector(f, bf16x2_t);
    return *(const uint32_t*)&v;
}
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    __device__ static inline void unpack(const uint4& u, float* f) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    __device__ static inline uint4 pack(const float* f) {
        return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]),
                          pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
    }
};

// Physical 16-byte slot of logical slot L in row `row` of an LDS operand slab with ns slots per row
// (ns = 4, 8, 16, 32 ...), read by MFMA A-fragment loads: lane (fr, fg) reads slot kc*4 + fg of row
// r0 + fr with ds_read_b128.  The LDS serves a wave's b128 read in four groups of 16 lanes, and each
// group holds the eight lanes of an EVEN fg whose rows form one cyclic run of 8 plus the eight lanes of
// the next ODD fg on the other 8 rows (MI355X_MICROARCH.md, LDS table).  A plain XOR-with-row swizzle
// lets the two halves collide whenever the run starts on an odd row (every other conv tap): measured
// SQ_LDS_BANK_CONFLICT = 26-47 % of the LDS cycles.  Here the parity of L picks the half of the 256-byte
// bank line, so even-fg and odd-fg lanes can never meet, and inside a half the remaining slot bits are
// XOR-ed with the row: 8 consecutive rows -> 8 distinct places.  Conflict-free for every start row
// (tools/probes/lds_swizzle_sim.py); 8-lane contiguous stores stay conflict-free too.
struct SlabSwizzle {
    int nbm, half, sh;  // (slots per 256-byte bank line) - 1, half of them, log2(rows per bank line)
    __device__ explicit SlabSwizzle(int ns) {
        const int nb = ns >= 16 ? 16 : ns;
        nbm = nb - 1;
        half = nb >> 1;
        sh = ns >= 16 ? 0 : (ns == 8 ? 1 : 2);
    }
    __device__ inline int slot(int L, int row) const {
        return (L & ~nbm) | ((L & 1) * half) | ((((L & nbm) >> 1) ^ (row >> sh)) & (half - 1));
    }
    // the logical slot stored at physical slot ps of `row` (for LDS-DMA fills, which write lane l at l * 16)
    __device__ inline int logical(int ps, int row) const {
        return (ps & ~nbm) | ((((ps & (half - 1)) ^ ((row >> sh) & (half - 1))) << 1) | ((ps & half) ? 1 : 0));
    }
};

// Sum over the four lanes {l, l^16, l^32, l^48} (the four 16-lane groups that share an MFMA column)
// with v_permlane16_swap / v_permlane32_swap (VALU, gfx950) instead of two ds_bpermute round trips:
// swapping the odd rows of one copy with the even rows of another leaves {own, partner} in the two
// results.  Same association as  x += shfl_xor(x, 16); x += shfl_xor(x, 32).
__device__ inline float group4_sum(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
#else
    return x;
#endif
}

// Whole-wave reductions without LDS round trips: four DPP steps inside each row of 16 lanes (xor 1, xor 2, half-row mirror,
// row mirror), then the four rows through v_permlane16_swap / v_permlane32_swap.  (r03: the __shfl_xor butterfly these
// replace compiles to six dependent ds_bpermute round trips per reduction; the row kernels built on it - LayerNorm forward /
// backward, softmax, row dots - spent their time in that chain, not on memory: with four rows in flight per wave and a
// quarter of the waves they got SLOWER.)  Every lane receives the result.
__device__ inline float row16_sum(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false));  // row_half_mirror
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xf, 0xf, false));  // row_mirror
#endif
    return v;
}
__device__ inline float row16_max(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xf, 0xf, false)));
#endif
    return v;
}
__device__ inline float wave_sum(float v) { return group4_sum(row16_sum(v)); }
__device__ inline float wave_max(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    v = row16_max(v);
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
#else
    return v;
#endif
}

// ---- MFMA wrappers: one "16-byte K chunk per lane" step -------------------------------------
// Both operands are read K-contiguously, 16 bytes per lane.  For bf16 that is one MFMA
// (8 k-values per lane); for fp32 it is four MFMAs (one k-value per lane each).  Because the same
// (lane-group, element) -> k assignment is used for both operands, the k order inside a chunk
// is a free permutation of the dot product.
//
// 16x16 tile: "row operand" lane supplies R[i = lane&15][k], "col operand" C[k][j = lane&15];
// D: lane holds col j = lane&15, rows i = (lane>>4)*4 + reg.
template <typename T> struct Mma16;
template <> struct Mma16<bf16> {
    static constexpr int K_PER_CHUNK = 32;  // k-values one 16-B-per-lane step covers (4 groups x 8)
    __device__ static inline void step(const uint4& r, const uint4& c, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&r, *(const bf16x8_t*)&c, acc, 0, 0, 0);
    }
};
template <> struct Mma16<float> {
    static constexpr int K_PER_CHUNK = 16;  // 4 groups x 4 floats
    __device__ static inline void step(const uint4& r, const uint4& c, f32x4_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.x), __uint_as_float(c.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.y), __uint_as_float(c.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.z), __uint_as_float(c.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(r.w), __uint_as_float(c.w), acc, 0, 0, 0);
    }
};

// 32x32 tile: row operand R[i = lane&31][k], col operand C[k][j = lane&31], k-slots owned by
// hi = lane>>5; D: lane holds col j = lane&31, rows i = (reg&3) + 8*(reg>>2) + 4*hi.
template <typename T> struct Mma32;
template <> struct Mma32<bf16> {
    static constexpr int K_PER_CHUNK = 16;  // 2 groups x 8
    __device__ static inline void step(const uint4& r, const uint4& c, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8_t*)&r, *(const bf16x8_t*)&c, acc, 0, 0, 0);
    }
};
template <> struct Mma32<float> {
    static constexpr int K_PER_CHUNK = 8;  // 2 groups x 4
    __device__ static inline void step(const uint4& r, const uint4& c, f32x16_t& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r.x), __uint_as_float(c.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r.y), __uint_as_float(c.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r.z), __uint_as_float(c.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(r.w), __uint_as_float(c.w), acc, 0, 0, 0);
    }
};

// Eight fp32 values (two 16-byte chunks of a lane's K slice) -> bf16 heads and bf16 tails, x = hi + lo (+ 2^-17 |x|):
// hi = RNE(x) by v_cvt_pk_bf16_f32, lo = RNE(x - hi) with the subtraction exact in fp32.
__device__ inline void split_bf16x3(const uint4& c0, const uint4& c1, uint4& hi, uint4& lo) {
    const float f[8] = {__uint_as_float(c0.x), __uint_as_float(c0.y), __uint_as_float(c0.z), __uint_as_float(c0.w),
                        __uint_as_float(c1.x), __uint_as_float(c1.y), __uint_as_float(c1.z), __uint_as_float(c1.w)};
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = pack_bf16x2(f[2 * j], f[2 * j + 1]);
        const float r0 = f[2 * j] - __uint_as_float(h[j] << 16);
        const float r1 = f[2 * j + 1] - __uint_as_float(h[j] & 0xffff0000u);
        l[j] = pack_bf16x2(r0, r1);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// nn.Dropout mask bits of element i at a site: counter-based, 24 bits; an element is dropped when bits < p * 2^24.
// Regenerated wherever the mask is needed, never stored.  The site constant is a full splitmix64 of (seed, site key) - uniform
// in a launch, so it runs on the scalar unit; per element a 32-bit integer finaliser (xorshift-multiply, two rounds) of the
// index folded to 32 bits XOR that constant.  (The first version ran splitmix64 per element: three 64-bit multiplies = a dozen
// quarter-rate v_mul_lo/hi_u32 - the attention backward with dropout 0.1 spent more cycles hashing than multiplying, and the
// stand-alone dropout pass was VALU- instead of HBM-bound: 73 us for a 100 MB tensor.)
__host__ __device__ inline uint32_t dropout_bits(uint64_t seed, uint64_t key, uint64_t i) {
    uint64_t z = seed + key * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const uint32_t hi = (uint32_t)(i >> 32);
    uint32_t x = (uint32_t)i ^ (uint32_t)z ^ ((hi << 13) | (hi >> 19)) ^ (uint32_t)(z >> 32);
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x >> 8;
}

// LDS-DMA completion is made explicit wherever a barrier publishes DMA'd operands: hipcc usually
// puts a vmcnt(0) in front of such a barrier itself, but it may hoist that wait out of a loop (seen
// when VGPR-returning loads sit ahead of the loop), which leaves the back-edge barrier unprotected
// - a sporadic stale-operand race.
__device__ inline void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace fs2
