// Persistent form of the slab GEMM / implicit-GEMM conv (gemm_mfma.hip) for launches of MANY tiles per CU: bf16 in, bf16 out,
// plain (bias [+ ReLU]) or deferred-LayerNorm epilogue.
//
// Why (r04 / r05 measurements, profiles/HISTORY.md §4): the K = 768 GEMMs of the LightSpeech configs (M = 49152; 768 .. 2304 tiles of 12
// K-steps each on 256 CUs) ran at 0.95-1.0 PFLOP/s where the same K loop reaches 1.39 on the 36-step decoder conv - per tile a
// workgroup pays its launch, the round trip of its first operand DMAs, a bias fetch, 128 KB of stores draining with nothing else
// in flight and its teardown: ~7 k of ~44 k cycles.  Here ONE workgroup per CU walks its tiles:
//   * the K steps of consecutive tiles form ONE operand stream: the first step of tile i + 1 (activation slab, weight tile, bias
//     row) is requested at the top of the LAST step of tile i, into the buffers that step does not read;
//   * the epilogue's stores are left in flight: the wait in front of tile i + 1's first step is COUNTED (vmcnt = the stores this
//     wave issued behind its DMAs; vector-memory operations retire in issue order on gfx9), the barrier is the raw s_barrier;
//   * the bias row is fetched once per tile by every wave (one 16-byte load) and published in LDS by wave 7 - which requests no
//     DMA, so the wait in front of its ds_write concerns its own queue only - behind the first step's MFMAs; the slab kernel's
//     epilogue fetches it with sixteen conditional loads per lane behind the K loop, each a round trip of its own.
// Tile order: workgroup b runs on XCD b % 8; every XCD owns a contiguous range of tiles (column tile fastest) and its workgroups
// take them round-robin, so the tiles in flight on an XCD at any time are consecutive - the column tiles of a row tile share
// its activation slab in that XCD's L2, as in the one-tile-per-workgroup launch.
//
// Arithmetic per output element = the slab kernel's, operation for operation: the same LDS images and fragment maps, the same
// MFMA (v_mfma_f32_16x16x32_bf16, weights as the first operand) in the same order (channel block outer, tap inner, two 32-k
// chunks per step), the same epilogue expressions.  Bit-identical (tests/test_gpu_ops.py::
// test_persistent_gemm_is_bit_identical_to_the_slab_kernel), so which form a launch takes is a performance choice only.
#include <hip/hip_runtime.h>

#include "fs2_common.h"
#include "fs2_kernels.h"

namespace fs2 {
namespace {

constexpr int PR_ROWB = 128, PR_BN = 256, PR_KE = 64;
static_assert(PR_KE == 64, "K step");
__device__ __attribute__((unused)) inline int pr_swz(int row, int slot) { return row * PR_ROWB + ((slot ^ (row & 7)) << 4); }  // = gemm_mfma.hip's swz
__device__ __attribute__((unused)) inline int pr_wcol(int ni, int fgq) { return (ni >> 1) * 32 + fgq * 8 + (ni & 1) * 4; }    // = wcol
__device__ __attribute__((unused)) inline int pr_wswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }         // = wswz

template <bool B> struct BoolCP { static constexpr bool value = B; };
template <int I> struct IntCP { static constexpr int value = I; };

#define PR_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))

// Pointwise launches only (taps == 1): the slab holds the tile's own rows, no conv halo.  (A conv form with the tap loops compiled,
// spilled at every tile height - hipcc's allocation around the tap loops - and a spill reload in these loops is a vmcnt wait that
// drains the operand requests: dropped; dense convs stay on the slab kernel, 36 steps per tile.)
// RS: the row-scaled product (GemmArgs::rs_stats; plain epilogue only): per-row (rstd, rstd * mean) and the wg row reach the epilogue
// through LDS like the bias row - fetched at the top of the tile, published by waves 4-7 behind the first step's MFMAs.
// HEAD (deferred epilogue only; GemmArgs::head_out): the tile's rows are NOT stored - the consumer is a Linear(N, 1) head behind the
// LayerNorm, and sum_n LN(v)[n] w[n] = rstd (sum_n v[n] gw[n] - mean sum_n gw[n]) + const with gw = gamma * w.  The epilogue leaves
// the row statistics as always plus sum_n v[n] gw[n] over the tile's columns; launch_head_finish makes the prediction of them.
template <int MI, bool DEFER, bool RS = false, bool HEAD = false>
__global__ __launch_bounds__(512) void gemm_persist_kernel(GemmArgs p, int ntiles) {
    static_assert(!(RS && DEFER), "the row-scaled product is a plain-epilogue form");
    static_assert(!HEAD || DEFER, "the head sums ride the deferred-LayerNorm epilogue");
#if defined(__HIP_DEVICE_COMPILE__)
    using T = bf16;
    constexpr int BMs = MI * 32;
    constexpr int GROUPS = BMs / 8;  // 8-row (1 KiB) DMA groups
    constexpr int SI = (GROUPS + 7) / 8;
    constexpr int SLAB_B = SI * 8 * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char slab0[SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char slab1[SLAB_B];
    __shared__ __attribute__((aligned(16))) unsigned char wt0[PR_BN * PR_ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char wt1[PR_BN * PR_ROWB];
    __shared__ __attribute__((aligned(16))) float sbias[PR_BN];                 // the tile's bias row (wave 7 brings it in under the tile's first step)
    __shared__ __attribute__((aligned(16))) float swg[RS || HEAD ? PR_BN : 4];  // RS: sum_k W'[n][k] of the tile's columns; HEAD: gamma[n] * w_head[n]
    __shared__ __attribute__((aligned(16))) float red3[HEAD ? 4 * BMs : 4];     // HEAD: the head sums' exchange
    __shared__ __attribute__((aligned(16))) float2 srow[RS ? BMs : 2];          // RS: (rstd, rstd * mean) of the tile's rows
    __shared__ __attribute__((aligned(16))) float2 red[DEFER ? 4 * BMs : 2];    // row statistics exchange (its own object: the operand buffers
                                                                                // already hold the next tile's first step by then)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = p.S, ncc = p.Cin / PR_KE;  // ncc is even (the launcher checks): a tile's last step reads slab1 / wt1
    const int tiles_n = (p.N + PR_BN - 1) / PR_BN, tiles_m = (S + BMs - 1) / BMs;
    // ---- this workgroup's tiles
    const int slots = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int per = ntiles >> 3, rem = ntiles & 7;
    const int xbeg = xcd * per + (xcd < rem ? xcd : rem), xcnt = per + (xcd < rem ? 1 : 0);
    if (slot >= xcnt) return;
    auto decode = [&](int ti, int& t0, int& n0, int& ub, int& bn) {
        int r = xbeg + ti;
        bn = r % tiles_n; r /= tiles_n;
        const int tm = r % tiles_m;
        ub = r / tiles_m;
        t0 = tm * BMs;
        n0 = bn * PR_BN;
    };
    // ---- operand DMA (buffer loads straight into LDS; out-of-range lanes read zeros).  ONE descriptor over the whole activation
    // tensor (the launcher checks M * ldx * 2 < 4 GiB): the utterance offset lives in the per-lane offsets
    constexpr unsigned OOB = 0xFFFFF000u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, (unsigned)(((size_t)(p.M - 1) * p.ldx + p.Cin) * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (unsigned)((size_t)p.N * p.K * sizeof(T)), 0x00020000);
    constexpr int DW = 4, DSI = SI * 8 / DW, DWI = 32 / DW;  // the four older waves issue the DMAs (gemm_mfma.hip: they have the slack)
    const bool dma_wave = wave < DW;
    // Per-lane DMA offsets: piece i of a wave covers tile rows i * 32 + (wave & 3) * 8 + (lane >> 3), so its offset is the first
    // piece's plus i * 32 rows, and both swizzles ((row & 7); wswz(row): bits 1, 3, 4 of the row) do not depend on i.  ONE register
    // per operand (+ the row for the bounds select) instead of DSI + DWI: at 256-row tiles sixteen stored offsets were the spills.
    unsigned sbase = 0, wbase = 0;
    int strow = 0, wnrow = 0;
    const unsigned sstride = 32u * (unsigned)(p.ldx * (int)sizeof(T)), wstride = 32u * (unsigned)(p.K * (int)sizeof(T));
    auto dma_setup = [&](int t0, int n0, int ub) {
        const int r0 = (wave & (DW - 1)) * 8 + (lane >> 3), ps = lane & 7;
        strow = t0 + r0;
        sbase = (unsigned)(ub * S + strow) * (unsigned)(p.ldx * (int)sizeof(T)) + (unsigned)((ps ^ (r0 & 7)) << 4);
        wnrow = n0 + r0;
        wbase = (unsigned)wnrow * (unsigned)(p.K * (int)sizeof(T)) + (unsigned)((ps ^ pr_wswz(r0)) << 4);
    };
    // One DMA instruction = one PIECE (1 KiB: 8 rows of a tile); a step's operands are NPIECE pieces per DMA wave: DWI of the weight
    // tile, then DSI of the slab, requested in a bunch in front of the MFMAs of the step before.  (Measured r05, knob builds: spread
    // one per row block over that step's MFMAs - the weight-resident kernel's way, where a piece has a whole tile to land - the
    // C3 GEMMs ran 5-20 % SLOWER: the last pieces are requested at the end of the step and the closing wait then sits out their whole
    // round trip.  Two per row block of the first 32-k chunk spilled 48 registers at 256-row tiles.)
    constexpr int NPIECE = DWI + DSI;
    auto piece = [&](int k, unsigned char* wdst, unsigned char* sdst, int cc) {  // k: a constant after unrolling
        if (k < DWI) {
            const unsigned v = wnrow + k * 32 < p.N ? wbase + (unsigned)k * wstride : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wdst + (k * DW + wave) * 1024), 16, v,
                                                     cc * PR_KE * (int)sizeof(T), 0, 0);
        } else if (k < NPIECE) {
            const int j = k - DWI;
            const unsigned v = strow + j * 32 < S ? sbase + (unsigned)j * sstride : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (__attribute__((address_space(3))) void*)(sdst + (j * DW + wave) * 1024), 16, v, cc * PR_ROWB, 0, 0);
        }
    };
    auto issue_all = [&](unsigned char* wdst, unsigned char* sdst, int cc) {
        if (!dma_wave) return;
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) piece(k, wdst, sdst, cc);
    };
    f32x4_t acc[4][MI];  // [ni][mi]
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int xrow0 = wm * (MI * 16) + fr;
    int woff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int wrow = wn * 64 + pr_wcol(i, fr >> 2) + (fr & 3);
            woff[i][ks] = wrow * PR_ROWB + (((ks * 4 + fg) ^ pr_wswz(wrow)) << 4);
        }
    // Per 32-k chunk: the four weight fragments, then row block by row block one activation fragment and its four MFMAs (the next
    // block's fragment requested one block ahead).  Every accumulator still receives its chunks in the slab kernel's order - the
    // order ACROSS accumulators is free - and 24 fragment registers are live instead of 48: at 256-row tiles the difference
    // between fitting the 256-register budget and spilling (a spill reload in these loops is a vmcnt wait that drains the requests).
    auto compute = [&](const unsigned char* sl, const unsigned char* wt) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 fw[4], fx[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) fw[i] = *(const uint4*)(wt + woff[i][ks]);
            fx[0] = *(const uint4*)(sl + pr_swz(xrow0, ks * 4 + fg));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                if (mi + 1 < MI) fx[(mi + 1) & 1] = *(const uint4*)(sl + pr_swz(xrow0 + (mi + 1) * 16, ks * 4 + fg));
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) Mma16<T>::step(fw[ni], fx[mi & 1], acc[ni][mi]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // One K step that is NOT a tile's last: request the next step's operands, multiply this one, wait for what was requested and
    // meet the other waves.  The tile's LAST step - always on slab1 / wt1, ncc being even - stands alone behind the loop: it requests
    // the NEXT tile's first operands and ends without wait or barrier; the epilogue runs under those requests and ends with the
    // (counted) wait.  (As one macro with a run-time "last" flag hipcc could not tell which path leaves a DMA into wt0 pending and
    // put a vmcnt(0) in front of every fragment read of wt0 - behind the requests just issued.)
#define PR_STEP(slab_cur, slab_nxt, w_cur, w_nxt, ccv, HOOK)                                           \
    {                                                                                                  \
        issue_all(w_nxt, slab_nxt, (ccv) + 1);                                                         \
        compute(slab_cur, w_cur);                                                                      \
        if (HOOK) bias_publish();                                                                      \
        PR_VMCNT(0);                                                                                   \
        __builtin_amdgcn_s_barrier();                                                                  \
    }
    // The tile's bias row: wave 7 (it requests no DMA: the vmcnt wait hipcc puts in front of the ds_write below concerns its own
    // queue only) fetches it with one ordinary 16-byte load per lane at the top of the tile and publishes it in LDS behind the
    // first step's MFMAs - a step after the request, in front of the first step's closing barrier.  (By LDS-DMA beside the operand
    // requests it made hipcc wait, in front of the first step's fragment reads, for most of the requests just issued: its
    // LDS-DMA scoreboard did not keep the sixth destination apart.)
    // (every wave issues the load - unconditional, from a clamped index, so that no select or phi sits between the load and its
    //  one use and pulls the wait forward; columns past N receive values nobody reads)
    float4 bq, wq;
    float2 rq;
    auto bias_publish = [&]() {
        if constexpr (RS) {
            if (wave < 4) return;
            const int row = (wave - 4) * 64 + lane;
            if (row < BMs) srow[row] = rq;
            if (wave == 7) {
                *(float4*)(sbias + lane * 4) = bq;
                *(float4*)(swg + lane * 4) = wq;
            }
        } else {
            if (wave != 7) return;
            *(float4*)(sbias + lane * 4) = bq;
            if constexpr (HEAD) *(float4*)(swg + lane * 4) = wq;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    int t0, n0, ub, bn;
    int ti = slot;
    decode(ti, t0, n0, ub, bn);
    dma_setup(t0, n0, ub);
    issue_all(wt0, slab0, 0);
    PR_VMCNT(0);  // (unconditional, here: under a run-time "first tile" flag inside the loop hipcc kept the path "prologue, no wait" alive and
                  //  put a vmcnt(2) - the prologue's two spill stores - in front of every tile's first fragment reads, behind the step's requests)
    constexpr int NSTORE = 2 * MI;  // store instructions of a full tile's epilogue per wave
    for (;;) {
        const bool has_next = ti + slots < xcnt;
        int nt0 = 0, nn0 = 0, nub = 0, nbn = 0;
        if (has_next) decode(ti + slots, nt0, nn0, nub, nbn);
        // ---- accumulators: zero, or the residual (normalised on load if it is a pre-norm tensor) - gemm_mfma.hip's preload
        const bool res_in_acc = DEFER && !p.relu && p.epi_res != nullptr;
        // (lane-derived address pieces are laundered inside the tile loop: left alone hipcc hoists every tile-invariant per-lane
        //  offset of the preload and of the epilogue out of the loop and carries them - a dozen registers - across the K loop)
        int frp = fr, fgp = fg;
        asm volatile("" : "+v"(frp), "+v"(fgp));
        {
            // (literal zeros: hipcc folds them into the tile's first MFMAs as their C operand.  Laundering them through empty asm
            //  statements keeps the MFMAs tied to one register set but makes hipcc put a vmcnt(0) behind the tile's first barrier -
            //  in front of the first fragment reads, draining the previous tile's stores; with the 24-register fragment schedule
            //  below the renamed first step fits without spills in the loops)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < MI; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (DEFER) {
            if (res_in_acc) {
                const T* R = (const T*)p.epi_res + (size_t)ub * S * p.ldc;
                const bool rnorm = p.epi_res_stats != nullptr;
                const size_t rowbase0 = (size_t)ub * S;
                auto preload = [&](auto full_c) {
                    constexpr bool FULL = decltype(full_c)::value;
                    float rmean[MI], rrstd[MI];
                    int tr[MI];
#pragma unroll
                    for (int gi = 0; gi < MI; ++gi) {
                        const int t = t0 + wm * (MI * 16) + gi * 16 + frp;
                        tr[gi] = t >= S ? S - 1 : t;
                        rmean[gi] = 0.f;
                        rrstd[gi] = 1.f;
                    }
                    if (rnorm) {
                        float2 pq[MI][4];
#pragma unroll
                        for (int gi = 0; gi < MI; ++gi) {
                            const float2* ps = (const float2*)p.epi_res_stats + (rowbase0 + tr[gi]) * p.epi_res_parts;
#pragma unroll
                            for (int q = 0; q < 4; ++q) pq[gi][q] = ps[q < p.epi_res_parts ? q : 0];
                        }
                        const float invn = 1.0f / (float)p.N;
#pragma unroll
                        for (int gi = 0; gi < MI; ++gi) {
#pragma unroll
                            for (int q = 1; q < 4; ++q) if (q >= p.epi_res_parts) pq[gi][q] = make_float2(0.f, 0.f);
                            const float s1 = (pq[gi][0].x + pq[gi][1].x) + (pq[gi][2].x + pq[gi][3].x);
                            const float s2 = (pq[gi][0].y + pq[gi][1].y) + (pq[gi][2].y + pq[gi][3].y);
                            rmean[gi] = s1 * invn;
                            rrstd[gi] = 1.0f / sqrtf(fmaxf(__builtin_fmaf(-rmean[gi], rmean[gi], s2 * invn), 0.f) + p.ln_eps);
                        }
                    }
                    if constexpr (FULL) {
                        uint4 rq[MI][2];
#pragma unroll
                        for (int gi = 0; gi < MI; ++gi)
#pragma unroll
                            for (int j = 0; j < 2; ++j) rq[gi][j] = *(const uint4*)(R + (size_t)tr[gi] * p.ldc + n0 + wn * 64 + j * 32 + fgp * 8);
#pragma unroll
                        for (int gi = 0; gi < MI; ++gi)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const unsigned w4[4] = {rq[gi][j].x, rq[gi][j].y, rq[gi][j].z, rq[gi][j].w};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    acc[2 * j + (e >> 1)][gi][(2 * e) & 3] = __uint_as_float(w4[e] << 16);
                                    acc[2 * j + (e >> 1)][gi][(2 * e + 1) & 3] = __uint_as_float(w4[e] & 0xffff0000u);
                                }
                            }
                        if (rnorm) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int n = n0 + wn * 64 + j * 32 + fgp * 8;
#pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    const float gvr = p.epi_res_g[n + r], bvr = p.epi_res_b[n + r];
#pragma unroll
                                    for (int gi = 0; gi < MI; ++gi)
                                        acc[2 * j + (r >> 2)][gi][r & 3] = __builtin_fmaf((acc[2 * j + (r >> 2)][gi][r & 3] - rmean[gi]) * rrstd[gi], gvr, bvr);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int gi = 0; gi < MI; ++gi) {
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int n = n0 + wn * 64 + j * 32 + fgp * 8;
                                const T* src = R + (size_t)tr[gi] * p.ldc + n;
                                float rv[8];
#pragma unroll
                                for (int r = 0; r < 8; ++r) rv[r] = (n + r < p.N) ? Num<T>::to_f32(src[r]) : 0.f;
                                if (rnorm) {
#pragma unroll
                                    for (int r = 0; r < 8; ++r)
                                        rv[r] = n + r < p.N ? __builtin_fmaf((rv[r] - rmean[gi]) * rrstd[gi], p.epi_res_g[n + r], p.epi_res_b[n + r]) : 0.f;
                                }
#pragma unroll
                                for (int r = 0; r < 8; ++r) acc[2 * j + (r >> 2)][gi][r & 3] = rv[r];
                            }
                        }
                    }
                };
                if (n0 + PR_BN <= p.N) preload(BoolCP<true>{});
                else preload(BoolCP<false>{});
            }
        }
        // the per-lane DMA offsets of THIS tile once more (computed for its first requests inside the previous tile's last step):
        // recomputed from laundered inputs so that they are not live across the epilogue and the preload above (18 registers at
        // the kernel's pressure peak)
        if constexpr (DEFER) {
            int t0x = t0, n0x = n0, ubx = ub;
            asm volatile("" : "+s"(t0x), "+s"(n0x), "+s"(ubx));
            dma_setup(t0x, n0x, ubx);
        }
        // ---- the tile's first operands have landed (this wave's share: waited for at the end of the previous epilogue, or in the
        // prologue); everyone's after the barrier
        __builtin_amdgcn_s_barrier();
        {
            int nb = n0 + lane * 4;
            nb = nb < p.N - 4 ? nb : p.N - 4;
            bq = *(const float4*)(p.bias + nb);
            if constexpr (HEAD) wq = *(const float4*)(p.head_gw + nb);
            if constexpr (RS) {
                wq = *(const float4*)(p.rs_wg + nb);
                int rr = t0 + (wave & 3) * 64 + lane;
                rr = rr < S ? rr : S - 1;
                rq = ((const float2*)p.rs_stats)[(size_t)ub * S + rr];
            }
        }
        // the steps alternate between the two buffer pairs.  Step 0 stands outside the loop: it publishes the bias row, and with that
        // use of the pending load INSIDE the loop hipcc drains vmcnt in the loop's preheader - every wave's epilogue stores and the
        // bias fetch itself, a full round trip at the top of every tile
        PR_STEP(slab0, slab1, wt0, wt1, 0, true)
        for (int cc = 1; cc + 1 < ncc; cc += 2) {
            PR_STEP(slab1, slab0, wt1, wt0, cc, false)
            PR_STEP(slab0, slab1, wt0, wt1, cc + 1, false)
        }
        // ---- the tile's last step: its pieces are the next tile's first operands
        if (has_next) {
            dma_setup(nt0, nn0, nub);
            issue_all(wt0, slab0, 0);
        }
        compute(slab1, wt1);
        // ---- epilogue (under the next tile's first requests).  A tile that lies wholly inside (S, N) takes a path whose NSTORE
        // stores per wave are unconditional - the wait behind them can then be COUNTED (vmcnt(NSTORE): the requests above, older,
        // have landed; the stores stay in flight), and hipcc's own scoreboard can follow the count
        int fre = fr, fge = fg;
        asm volatile("" : "+v"(fre), "+v"(fge));
        const bool fulln = n0 + PR_BN <= p.N, fullt = t0 + BMs <= S;
        T* __restrict__ C = (T*)p.C + (size_t)ub * S * p.ldc;
        float bv[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 b0 = *(const float4*)(sbias + wn * 64 + j * 32 + fge * 8), b1 = *(const float4*)(sbias + wn * 64 + j * 32 + fge * 8 + 4);
            bv[j][0] = b0.x; bv[j][1] = b0.y; bv[j][2] = b0.z; bv[j][3] = b0.w;
            bv[j][4] = b1.x; bv[j][5] = b1.y; bv[j][6] = b1.z; bv[j][7] = b1.w;
        }
        if constexpr (!DEFER) {
            float rsr[RS ? MI : 1], rsm[RS ? MI : 1], wgv[RS ? 2 : 1][8];
            if constexpr (RS) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float2 q = srow[wm * (MI * 16) + mi * 16 + fre];
                    rsr[mi] = q.x;
                    rsm[mi] = q.y;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 w0 = *(const float4*)(swg + wn * 64 + j * 32 + fge * 8), w1 = *(const float4*)(swg + wn * 64 + j * 32 + fge * 8 + 4);
                    wgv[j][0] = w0.x; wgv[j][1] = w0.y; wgv[j][2] = w0.z; wgv[j][3] = w0.w;
                    wgv[j][4] = w1.x; wgv[j][5] = w1.y; wgv[j][6] = w1.z; wgv[j][7] = w1.w;
                }
            }
            auto store = [&](auto relu_c, auto full_c, auto rows_c) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int t = t0 + wm * (MI * 16) + mi * 16 + fre;
                    if (!decltype(rows_c)::value && t >= S) continue;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = n0 + wn * 64 + j * 32 + fge * 8;
                        if (!decltype(full_c)::value && n >= p.N) continue;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if constexpr (RS) v[r] = __builtin_fmaf(acc[2 * j + (r >> 2)][mi][r & 3], rsr[mi], __builtin_fmaf(-rsm[mi], wgv[j][r], bv[j][r]));
                            else v[r] = acc[2 * j + (r >> 2)][mi][r & 3] + bv[j][r];
                            if constexpr (decltype(relu_c)::value) v[r] = fmaxf(v[r], 0.f);
                        }
                        T* dst = (T*)((char*)C + (unsigned)(t * p.ldc + n) * (unsigned)sizeof(T));
                        if (decltype(full_c)::value || n + 7 < p.N) {
                            *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                        } else {
#pragma unroll
                            for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<T>::from_f32(v[r]);
                        }
                    }
                }
            };
            if (fulln && fullt) {
                if (p.relu) store(BoolCP<true>{}, BoolCP<true>{}, BoolCP<true>{});
                else store(BoolCP<false>{}, BoolCP<true>{}, BoolCP<true>{});
                PR_VMCNT(NSTORE);
            } else {
                if (fulln) {
                    if (p.relu) store(BoolCP<true>{}, BoolCP<true>{}, BoolCP<false>{});
                    else store(BoolCP<false>{}, BoolCP<true>{}, BoolCP<false>{});
                } else {
                    if (p.relu) store(BoolCP<true>{}, BoolCP<false>{}, BoolCP<false>{});
                    else store(BoolCP<false>{}, BoolCP<false>{}, BoolCP<false>{});
                }
                PR_VMCNT(0);
            }
        } else {
            // deferred-LayerNorm epilogue: v = act(acc + bias) [+ res, already in acc], stored as it is, + this column tile's
            // (sum v, sum v^2) per row -> stats_out[row][column tile]   (gemm_mfma.hip, DEFER)
            const size_t rowbase = (size_t)ub * S;
            const float lo = p.relu ? 0.f : -__builtin_inff();
            float a1[MI], a2[MI], a3[HEAD ? MI : 1];
            float gwv[HEAD ? 2 : 1][8];
            if constexpr (HEAD) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float4 w0 = *(const float4*)(swg + wn * 64 + j * 32 + fge * 8), w1 = *(const float4*)(swg + wn * 64 + j * 32 + fge * 8 + 4);
                    gwv[j][0] = w0.x; gwv[j][1] = w0.y; gwv[j][2] = w0.z; gwv[j][3] = w0.w;
                    gwv[j][4] = w1.x; gwv[j][5] = w1.y; gwv[j][6] = w1.z; gwv[j][7] = w1.w;
                }
            }
            auto body = [&](auto full_c, auto rows_c) {
                constexpr bool FULL = decltype(full_c)::value, ROWS = decltype(rows_c)::value;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const int t = t0 + wm * (MI * 16) + mi * 16 + fre;
                    a1[mi] = a2[mi] = 0.f;
                    if constexpr (HEAD) a3[mi] = 0.f;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int n = n0 + wn * 64 + j * 32 + fge * 8;
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(acc[2 * j + (r >> 2)][mi][r & 3] + bv[j][r], lo);
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if (!FULL && n + r >= p.N) v[r] = 0.f;
                            a1[mi] += v[r];
                            a2[mi] = __builtin_fmaf(v[r], v[r], a2[mi]);
                            if constexpr (HEAD) a3[mi] = __builtin_fmaf(v[r], gwv[j][r], a3[mi]);  // (columns past N: v = 0)
                        }
                        if (!HEAD && (ROWS || t < S) && (FULL || n < p.N)) {
                            T* dst = (T*)((char*)C + (unsigned)(t * p.ldc + n) * (unsigned)sizeof(T));
                            if (FULL || n + 7 < p.N) {
                                *(uint4*)dst = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                            } else {
#pragma unroll
                                for (int r = 0; r < 8; ++r) if (n + r < p.N) dst[r] = Num<T>::from_f32(v[r]);
                            }
                        }
                    }
                }
            };
            if (fulln && fullt) body(BoolCP<true>{}, BoolCP<true>{});
            else if (fulln) body(BoolCP<true>{}, BoolCP<false>{});
            else body(BoolCP<false>{}, BoolCP<false>{});
            if (p.stats_out) {  // the four column waves meet in LDS; the rows leave from waves 4 .. (not the DMA waves: their store count stays NSTORE)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float s1 = group4_sum(a1[mi]), s2 = group4_sum(a2[mi]);
                    if (fge == 0) red[wn * BMs + wm * (MI * 16) + mi * 16 + fre] = make_float2(s1, s2);
                    if constexpr (HEAD) {
                        const float s3 = group4_sum(a3[mi]);
                        if (fge == 0) red3[wn * BMs + wm * (MI * 16) + mi * 16 + fre] = s3;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int row = tid - 256;
                if (row >= 0 && row < BMs) {
                    const int t = t0 + row;
                    if (t < S) {
                        const float2 q0 = red[row], q1 = red[BMs + row], q2 = red[2 * BMs + row], q3 = red[3 * BMs + row];
                        ((float2*)p.stats_out)[(rowbase + t) * (size_t)tiles_n + bn] =
                            make_float2((q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y));
                        if constexpr (HEAD)
                            p.head_out[(rowbase + t) * (size_t)tiles_n + bn] = (red3[row] + red3[BMs + row]) + (red3[2 * BMs + row] + red3[3 * BMs + row]);
                    }
                }
            }
            // DMA waves (0-3): exactly NSTORE stores behind the requests on the full path; waves 4-7 requested nothing
            if (!HEAD && fulln && fullt && dma_wave) PR_VMCNT(NSTORE);
            else PR_VMCNT(0);
        }
        if (!has_next) break;
        ti += slots;
        t0 = nt0; n0 = nn0; ub = nub; bn = nbn;
    }
#undef PR_STEP
#else
    (void)p; (void)ntiles;
#endif
}

}  // namespace

// (Tuning::gemm_persist - knob 220 / 221: always one tile per workgroup / multi-round bf16 pointwise launches on this kernel (default))

// The tile height comes from the slab launcher's cost model (mi); this only says whether the persistent form can run the launch.
bool gemm_persist_supported(const GemmArgs& a, int in_dtype, int out_dtype, int mi) {
    if (in_dtype != FS2_BF16 || out_dtype != FS2_BF16) return false;
    if (a.ln_g || a.dot_w || a.z_out || a.res || a.gate || a.zero_rows || a.C_lo || a.split || a.w_presplit || a.ksplit > 1 || a.drop_p > 0.f) return false;
    if (a.rs_stats && (a.relu || a.stats_out || a.epi_res || !a.rs_wg)) return false;
    if (!a.bias) return false;
    // the kernel reads bias / rs_wg / head_gw as float4 (the slab kernel takes any 4-byte-aligned pointer): unaligned callers stay on it
    if (((uintptr_t)a.bias & 15) || ((uintptr_t)a.rs_wg & 15) || ((uintptr_t)a.head_gw & 15)) return false;
    if (a.head_out) {  // rows not stored: statistics + head sums only
        if (!a.head_gw || !a.stats_out || a.epi_res || a.rs_stats || mi != 6) return false;
    } else if (!a.C) {
        return false;
    }
    const bool defer = a.stats_out || a.epi_res;
    if (defer && (mi != 6 || (a.epi_res && a.relu))) return false;
    if (!defer && mi != 6 && mi != 8) return false;
    if (a.taps != 1 || a.Cin % 128 || a.K != a.taps * a.Cin || a.N < 192 || a.N % 8) return false;  // Cin / 64 even
    const int S = a.taps == 1 ? a.M : a.S;
    if (S <= 0 || a.M % S) return false;
    if (a.ldx % 8 || a.ldc % 8) return false;
    if ((size_t)a.M * a.ldx * 2 >= 0xFFFFF000ull || (size_t)S * a.ldc * 2 >= 0xFFFFF000ull || (size_t)a.N * a.K * 2 >= 0xFFFFF000ull) return false;
    return true;
}

bool gemm_head_supported(const GemmArgs& a, int in_dtype, int out_dtype) {
    return tuning_of(a.tune).head_sums && gemm_persist_supported(a, in_dtype, out_dtype, 6);  // (this kernel is the only form: independent of Tuning::gemm_persist)
}

static int persist_cus() {
    static int n = 0;  // one device model per process (MI355X: 256)
    if (!n) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        n = v & ~7;
    }
    return n;
}

// tiles per launch of the one-tile-per-workgroup form at this tile height; the persistent form pays from the second round on
bool gemm_persist_pays(const GemmArgs& a, int mi) {
    const int S = a.taps == 1 ? a.M : a.S, bm = mi * 32;
    const long tiles = (long)(a.M / S) * ((S + bm - 1) / bm) * ((a.N + PR_BN - 1) / PR_BN);
    return tiles > persist_cus();
}

template <int MI, bool DEFER, bool RS = false, bool HEAD = false>
static int launch_persist_t(const GemmArgs& a, hipStream_t stream) {
    const int BMs = MI * 32;
    const int tiles = (a.M / a.S) * ((a.S + BMs - 1) / BMs) * ((a.N + PR_BN - 1) / PR_BN);
    int grid = persist_cus();
    if (grid > ((tiles + 7) & ~7)) grid = (tiles + 7) & ~7;
    hipLaunchKernelGGL((gemm_persist_kernel<MI, DEFER, RS, HEAD>), dim3(grid), dim3(512), 0, stream, a, tiles);
    return hipGetLastError() == hipSuccess ? FS2_OK : FS2_ERR_HIP;
}

int launch_gemm_persist(const GemmArgs& a_in, int mi, hipStream_t stream) {
    GemmArgs a = a_in;
    if (a.taps == 1) a.S = a.M;
    const bool defer = a.stats_out || a.epi_res;
    if (defer && a.head_out) return launch_persist_t<6, true, false, true>(a, stream);
    if (defer) return launch_persist_t<6, true>(a, stream);
    if (a.rs_stats) return mi == 8 ? launch_persist_t<8, false, true>(a, stream) : launch_persist_t<6, false, true>(a, stream);
    if (mi == 8) return launch_persist_t<8, false>(a, stream);
    return launch_persist_t<6, false>(a, stream);
}

}  // namespace fs2
