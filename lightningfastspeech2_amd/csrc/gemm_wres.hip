// Weight-resident GEMM for K = 256 (the hidden size of the FS2-27M family): y = act(x W^T + b), bf16.
//
// The slab kernel (gemm_mfma.hip) re-streams a column tile's 256 x 256 weight block (128 KiB) through LDS for every row tile:
// the decoder's MHA in-projection (M = 49152, N = 768, K = 256) moves 672 KB per 128-row tile of which 512 KB are weights, and
// runs at a third of the rate its 100 MB of activations would allow (profiles/HISTORY.md §4, "what bounds the GEMM launches").  Here a
// workgroup (8 waves, 2 x 4 as in the slab kernel) loads its column tile's weights ONCE, straight from global memory into MFMA
// fragments - a wave's 64 channels x 256 k = 32 KiB = 128 VGPRs per lane - and then walks row tiles of 96 rows: the only LDS
// traffic is the activation tile (48 KiB, double-buffered, buffer-load-to-LDS DMA with the SlabSwizzle applied on the source
// side) whose DMA instructions are issued one per k-step between the MFMAs of the tile before it, one barrier per tile, and a
// counted wait at the top of a tile that leaves the previous tile's six stores in flight.
//
// Arithmetic per output element = the slab kernel's: fp32 accumulator from 0, the 256 k-values in ascending blocks of 32
// (v_mfma_f32_16x16x32_bf16, weights as the first operand), + bias, ReLU, one rounding to bf16.  Bit-identical to it
// (tests/test_gpu_ops.py::test_wres_gemm_is_bit_identical_to_the_slab_kernel), so which kernel a launch takes is a
// performance choice only.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {
namespace {

constexpr int WR_K = 256, WR_ROWB = WR_K * 2, WR_MI = 3, WR_RT = 2 * WR_MI * 16;  // 96 rows per tile, 48 per wave row
constexpr int WR_TILEB = WR_RT * WR_ROWB;                                        // 48 KiB
__attribute__((unused)) constexpr int WR_NDMA = WR_TILEB / 1024 / 8;                                     // 1-KiB DMA instructions per wave per tile

__device__ __attribute__((unused)) inline int wr_wcol(int ni, int fgq) { return (ni >> 1) * 32 + fgq * 8 + (ni & 1) * 4; }  // = gemm_mfma.hip's wcol

#ifdef FS2_WRES_PROBE  // tools/probes/wres_stamps.py: s_memtime of waves 0 and 7 of workgroup 0 around the phases of its first tiles
__device__ unsigned long long g_wres_stamps[2][64];
#define WRES_STAMP(i)                                                                                                      \
    do {                                                                                                                   \
        if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && (i) < 64)                                        \
            g_wres_stamps[threadIdx.x ? 1 : 0][i] = __builtin_amdgcn_s_memtime();                                           \
    } while (0)
#else
#define WRES_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(512) void gemm_wres_kernel(GemmArgs p, int ncol, int nrg, int tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    // two activation tiles (first: the weight staging area).  Separate objects: hipcc tracks pending LDS-DMA per object and would
    // wait for the next tile's DMA in front of every fragment read of this one if they were one array (measured: K loop x 4)
    __shared__ __attribute__((aligned(16))) unsigned char xs0[WR_TILEB];
    __shared__ __attribute__((aligned(16))) unsigned char xs1[WR_TILEB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fg = lane >> 4;
    // workgroup b runs on XCD b % 8: the ncol column-tile workgroups of one row group are consecutive on ONE XCD, so the row
    // group's activations are fetched into that XCD's L2 once
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int ct = q % ncol, rg = (q / ncol) * 8 + xcd;
    if (rg >= nrg) return;
    const int tbeg = (int)((long)rg * tiles / nrg), tend = (int)((long)(rg + 1) * tiles / nrg);
    if (tbeg >= tend) return;
    const int n0 = ct * 256;
    WRES_STAMP(0);

    // ---- this wave's weights: fragment (ks, ni) = channels n0 + wn*64 + wcol(ni, fr>>2) + (fr&3), k = ks*32 + fg*8 .. +7.
    // A lane's 16 bytes sit 512 bytes from its neighbour's (16 rows per instruction): loaded straight from global memory that is
    // 64 sectors per instruction and measured ~8 us for the 128 KiB.  They come in as the slab kernel's weight tiles instead -
    // whole 128-byte lines by LDS-DMA, [256 channels][64 k] per tile with its row swizzle - two tiles at a time.
    uint4 fw[8][4];
    {
        const __amdgpu_buffer_rsrc_t wrs =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * WR_K * 2), 0x00020000);
        auto wswz = [](int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };  // = gemm_mfma.hip's wswz
        auto stage = [&](int cc0) {  // weight tiles cc0 -> xs0, cc0 + 1 -> xs1
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // 32 x 1 KiB per tile, 4 per wave: 8 lanes = one channel's 128 bytes
                    const int P = (i * 8 + wave) * 64 + lane, row = P >> 3, ps = P & 7;
                    const unsigned voff = (unsigned)((n0 + row) * (WR_K * 2) + (cc0 + c) * 128 + ((ps ^ wswz(row)) << 4));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)((c ? xs1 : xs0) + (i * 8 + wave) * 1024),
                                                             16, voff, 0, 0, 0);
                }
            dma_drain();
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int wrow = wn * 64 + wr_wcol(ni, fr >> 2) + (fr & 3);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        fw[(cc0 + c) * 2 + h][ni] = *(const uint4*)((c ? xs1 : xs0) + wrow * 128 + (((h * 4 + fg) ^ wswz(wrow)) << 4));
                }
            __syncthreads();
        };
        stage(0);
        stage(2);
    }
    // the column tile's bias sits in LDS (sixteen registers per lane otherwise: with the weights resident the file is full, and a
    // spilled register's reload is a vmcnt(0) in the K loop)
    __shared__ __attribute__((aligned(16))) float sbias[256];
    if (tid < 256) sbias[tid] = p.bias ? p.bias[n0 + tid] : 0.f;
    __syncthreads();

    // ---- activation tile DMA: 2 rows (1 KiB) per instruction; physical slot ps of row i holds logical slot ls
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)(((size_t)(p.M - 1) * p.ldx + WR_K) * 2), 0x00020000);
    // buffers are named by compile-time index everywhere (no pointer selects): hipcc then knows which LDS object a fragment read
    // touches and does not wait for the DMA just issued into the OTHER one
    // An issue is never under a branch: at a control-flow join hipcc's LDS-DMA scoreboard turns "maybe pending" into a vmcnt(0)
    // in front of every later DMA and fragment read (seen in the ISA; the K loop ran 4 x longer).  `live` = false sends every
    // lane out of range instead (no memory traffic, zeros written).
    auto issue_piece = [&](auto BUF, int t, int k, bool live) {
        unsigned char* const dst = decltype(BUF)::value ? xs1 : xs0;
        const int c = k * 8 + wave, i = 2 * c + (lane >> 5), ps = lane & 31, m = t * WR_RT + i;
        const int ls = (ps & 16) | ((((ps & 7) ^ (i & 7)) << 1) | ((ps >> 3) & 1));
        const unsigned voff = (live && m < p.M) ? (unsigned)m * (unsigned)(p.ldx * 2) + (unsigned)(ls << 4) : 0xFFFFF000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, voff, 0, 0, 0);
    };
    auto issue = [&](auto BUF, int t) {
#pragma unroll
        for (int k = 0; k < WR_NDMA; ++k) issue_piece(BUF, t, k, true);
    };
    // fragment (mi, ks) of this lane: row wm*48 + mi*16 + fr (row & 7 = fr & 7), slot ks*4 + fg -> SlabSwizzle(32)::slot
    const int arow = (wm * (WR_MI * 16) + fr) * WR_ROWB;
    const int acx = (((fg & 1) << 3) | ((fg >> 1) ^ (fr & 7))) << 4;
    // side(ks): one DMA instruction of the next tile and one 16-byte store of the previous tile's rows per k-step, BETWEEN the
    // MFMAs - issued in a bunch behind the barrier they took 2-3.5 k of a tile's 8 k ticks (eight waves, one address unit, every
    // wave in the same phase; tools/probes/wres_stamps.py)
    auto compute = [&](auto BUF, f32x4_t (&acc)[4][WR_MI], auto side) {
        const unsigned char* const xs = decltype(BUF)::value ? xs1 : xs0;
        // a k-step's three fragments are requested before the MFMAs of the step before it: with two waves per SIMD both in their
        // K loops an LDS round trip per step is otherwise exposed (MFMA pipe 0.32 busy without)
        uint4 fx[2][WR_MI];
        auto ldx = [&](uint4 (&f)[WR_MI], int ks) {
#pragma unroll
            for (int mi = 0; mi < WR_MI; ++mi)
                f[mi] = *(const uint4*)(xs + arow + mi * 16 * WR_ROWB + (acx ^ ((((ks & 3) << 1) | ((ks >> 2) << 4)) << 4)));
        };
        ldx(fx[0], 0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 1 < 8) ldx(fx[(ks + 1) & 1], ks + 1);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
                for (int mi = 0; mi < WR_MI; ++mi) Mma16<bf16>::step(fw[ks][ni], fx[ks & 1][mi], acc[ni][mi]);
                if (ni == 0) side(ks);
            }
        }
    };
    auto store_piece = [&](int t, int mi, int j, const uint4& q) {
        const int m = t * WR_RT + wm * (WR_MI * 16) + mi * 16 + fr;
        if (m < p.M) *(uint4*)((bf16*)p.C + (size_t)m * p.ldc + n0 + wn * 64 + j * 32 + fg * 8) = q;
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    WRES_STAMP(1);
    issue(B0{}, tbeg);
    auto tile = [&](auto CUR, auto NXT, int t) {
        WRES_STAMP(4 + (t - tbeg) * 5);
        // this wave's share of tile t has landed.  Its DMAs were issued inside tile t-1's K loop, BEFORE that tile's six stores
        // (every tile with a successor lies wholly inside M, so all six were issued): vmcnt(6) leaves only those in flight -
        // vector-memory operations of one kind complete in issue order on gfx9, the compiler's own counted waits rely on it
        if (t == tbeg) __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0)
        else __builtin_amdgcn_s_waitcnt(0x0F70 | 6);                  // vmcnt(6)
        WRES_STAMP(5 + (t - tbeg) * 5);
        __syncthreads();  // everyone's share; every wave has left tile t-1's K loop: the other buffer is free
        WRES_STAMP(6 + (t - tbeg) * 5);
        const bool has_next = t + 1 < tend;
        auto side = [&](int ks) {
            if (ks < WR_NDMA) issue_piece(NXT, t + 1, ks, has_next);
        };
        WRES_STAMP(7 + (t - tbeg) * 5);
        f32x4_t acc[4][WR_MI];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < WR_MI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        compute(CUR, acc, side);
        WRES_STAMP(8 + (t - tbeg) * 5);
#pragma unroll
        for (int mi = 0; mi < WR_MI; ++mi) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 b0 = *(const float4*)(sbias + wn * 64 + j * 32 + fg * 8), b1 = *(const float4*)(sbias + wn * 64 + j * 32 + fg * 8 + 4);
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    v[r] = acc[2 * j + (r >> 2)][mi][r & 3] + bv[r];
                    if (p.relu) v[r] = fmaxf(v[r], 0.f);
                }
                store_piece(t, mi, j, make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])));
            }
        }
    };
    for (int t = tbeg; t < tend; t += 2) {
        tile(B0{}, B1{}, t);
        if (t + 1 < tend) tile(B1{}, B0{}, t + 1);
    }
#else
    (void)p; (void)ncol; (void)nrg; (void)tiles;
#endif
}

}  // namespace

// (Tuning::gemm_wres - 1: bf16 K = 256 plain GEMMs take this kernel where it pays; 2: wherever it applies (tests); 0: never)

static int wres_row_groups(int ncol, int tiles) {
    int nrg = (256 / ncol) & ~7;  // row groups: a multiple of 8 (one per XCD and round), every CU at most one workgroup
    if (nrg < 8) nrg = 8;
    if (nrg > ((tiles + 7) & ~7)) nrg = (tiles + 7) & ~7;
    return nrg;
}

// force = false: also ask whether it PAYS - a workgroup spends ~3 us bringing its weights in, which it earns back from its third
// row tile on (measured r03, C2: decoder in-projection 49152 x 768: 28.3 us against 33.6-34.3 on the slab kernel; the encoder's
// 8192 x 768, one tile per workgroup: 12.6 against 10.9-11.8).  Results are bit-identical either way.
bool gemm_wres_supported(const GemmArgs& a, int in_dtype, int out_dtype, bool force) {
    if (a.N >= 256 && a.N % 256 == 0 && !force) {
        const int tiles = (a.M + WR_RT - 1) / WR_RT;
        if (tiles < 3 * wres_row_groups(a.N / 256, tiles)) return false;
    }
    return in_dtype == FS2_BF16 && out_dtype == FS2_BF16 && a.taps == 1 && a.K == WR_K && a.Cin == WR_K && a.N >= 256 && a.N % 256 == 0 &&
           a.N <= 2048 && a.M > 0 && !a.res && !a.ln_g && !a.dot_w && !a.z_out && !a.epi_res && !a.stats_out && !a.gate && !a.zero_rows &&
           !a.split && !a.rs_stats && a.ldx % 8 == 0 && a.ldc % 8 == 0 && (size_t)a.M * a.ldx * 2 < 0xFFFFF000ull && (size_t)a.M * a.ldc * 2 < 0xFFFFF000ull;
}

int launch_gemm_wres(const GemmArgs& a, hipStream_t stream) {
    const int ncol = a.N / 256, tiles = (a.M + WR_RT - 1) / WR_RT, nrg = wres_row_groups(ncol, tiles);
    hipLaunchKernelGGL(gemm_wres_kernel, dim3((unsigned)(nrg * ncol)), dim3(512), 0, stream, a, ncol, nrg, tiles);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

}  // namespace fs2

#ifdef FS2_WRES_PROBE
extern "C" int fs2_dbg_wres_stamps(unsigned long long* out /*2 x 64, host*/) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fs2::g_wres_stamps), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
}
#endif
