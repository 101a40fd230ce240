# derive the stamped probe sources from the current csrc/ kernels -> /tmp/fs2_stamp/*.hip
import os
import re
os.makedirs("/tmp/fs2_stamp", exist_ok=True)
R='/root/repo/lightningfastspeech2_amd/csrc/'
rb=open(R+'vocoder_resblock.hip').read()
def rep(s,a,b,n=1):
    assert s.count(a)>=1,a[:60]
    return s.replace(a,b,n)
rb=rep(rb,"namespace fs2 {\n\nnamespace {\ntemplate <typename T> struct RbT;",'''namespace fs2 {
__device__ unsigned long long g_rb_stamps[48 * 8 * 16];
#define STAMP_AT(k) do { if (st_ptr) st_ptr[(k)] = __builtin_amdgcn_s_memtime(); } while (0)

namespace {
template <typename T> struct RbT;''')
rb=rep(rb,"    const int tbase = t0 - H;  // time of tile row 0; slab index i <-> tile row i - G\n",'''    const int tbase = t0 - H;  // time of tile row 0; slab index i <-> tile row i - G
    unsigned long long* st_ptr = nullptr;
    unsigned long long* st_w = nullptr;  // pairs: every wave's end of K loop 0
    {
        const int per = gridDim.x / 8 > 0 ? gridDim.x / 8 : 1;
        const int ci = ((p.C == 32 ? 0 : (p.C == 64 ? 1 : 2)) * 3 + (p.taps - 3) / 4) * 4 + (p.npairs == 3 ? 3 : (p.dil[0] - 1) / 2);
        if (tid == 0 && (int)blockIdx.x % per == per / 2 && (int)blockIdx.x / per < 8) st_ptr = g_rb_stamps + (ci * 8 + blockIdx.x / per) * 16;
        if (p.npairs == 1 && lane == 0 && (int)blockIdx.x % per == per / 2 && (int)blockIdx.x / per < 8) st_w = g_rb_stamps + (ci * 8 + blockIdx.x / per) * 16 + 6 + wave;
    }
    STAMP_AT(0);
    if (st_ptr) st_ptr[15] = wall_clock64();
''')
rb=rep(rb,"    if (p.x_act) dma_drain();  // this wave's slab DMAs have landed before the barrier publishes them\n    __syncthreads();\n","    if (p.x_act) dma_drain();\n    __syncthreads();\n    STAMP_AT(1);\n")
rb=rep(rb,"        // ---- epilogue: lane = rows (m*16 + fr), channels n0 .. n0+7 ----\n","        STAMP_AT(2 + 2 * j);\n        if (st_w && j == 0) *st_w = __builtin_amdgcn_s_memtime();\n        // ---- epilogue: lane = rows (m*16 + fr), channels n0 .. n0+7 ----\n")
rb=rep(rb,"        if (!last) __syncthreads();\n    }\n}","        if (!last) __syncthreads();\n        STAMP_AT(3 + 2 * j);\n    }\n    if (st_ptr) st_ptr[14] = wall_clock64();\n}")
rb=rep(rb,"int g_voc_fused_resblock = 1;",'''}  // namespace fs2
extern "C" int fs2_dbg_rb_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs2::g_rb_stamps), sizeof(unsigned long long) * 48 * 8 * 16);
}
namespace fs2 {
int g_voc_fused_resblock = 1;''')
rb=rep(rb,'dim3(NW * 64), smem, stream,','dim3(NW * 64), smem + (NW == 4 ? SOLO_PAD : 0), stream,')
rb=rb.replace('namespace fs2 {\n__device__ unsigned long long g_rb_stamps','#ifndef SOLO_PAD\n#define SOLO_PAD 0  // -DSOLO_PAD=24576: one 4-wave workgroup per CU (solo K-loop rate)\n#endif\nnamespace fs2 {\n__device__ unsigned long long g_rb_stamps',1)
open('/tmp/fs2_stamp/vocoder_resblock.hip','w').write(rb)
cv=open(R+'vocoder_conv.hip').read()
cv=rep(cv,"namespace fs2 {\n\nnamespace {\n\ntemplate <typename T> struct VocT;",'''namespace fs2 {
__device__ unsigned long long g_cv_stamps[64 * 8 * 8];
#define STAMP_AT(k) do { if (st_ptr) st_ptr[(k)] = __builtin_amdgcn_s_memtime(); } while (0)

namespace {

template <typename T> struct VocT;''')
cv=rep(cv,"    if (t0 >= len) return;  // block-uniform\n",'''    if (t0 >= len) return;  // block-uniform
    unsigned long long* st_ptr = nullptr;
    {
        const int per = gridDim.x / 8 > 0 ? gridDim.x / 8 : 1;
        const int ci = ((p.cin_pad >> 5) * 7 + p.taps * 3 + p.dil + (p.n >> 5) * 5 + (p.res ? 1 : 0)) & 63;
        if (tid == 0 && blockIdx.y == 0 && (int)blockIdx.x % per == per / 2 && (int)blockIdx.x / per < 8) {
            st_ptr = g_cv_stamps + (ci * 8 + blockIdx.x / per) * 8;
            st_ptr[6] = ((unsigned long long)p.cin_pad << 48) | ((unsigned long long)p.n << 32) | ((unsigned long long)p.taps << 24) | ((unsigned long long)p.dil << 16) | (p.res ? 2 : 0) | (p.accumulate ? 1 : 0) | ((unsigned long long)MI16 << 8);
        }
    }
    STAMP_AT(0);
    if (st_ptr) st_ptr[5] = wall_clock64();
''')
cv=rep(cv,"    __syncthreads();\n\n    const int wrow0 = wm * RW;\n","    __syncthreads();\n    STAMP_AT(1);\n\n    const int wrow0 = wm * RW;\n")
cv=rep(cv,"    // ---- epilogue: lane = rows (m*16 + fr), 8 consecutive channels n0 .. n0+7 ----\n","    STAMP_AT(2);\n    // ---- epilogue: lane = rows (m*16 + fr), 8 consecutive channels n0 .. n0+7 ----\n")
cv=rep(cv,'''                for (int q = 0; q < NP; ++q) dst[q] = Vec16<T>::pack(v + q * E16);
            }
        }
    }
}
''','''                for (int q = 0; q < NP; ++q) dst[q] = Vec16<T>::pack(v + q * E16);
            }
        }
    }
    STAMP_AT(3);
    if (st_ptr) st_ptr[4] = wall_clock64();
}
''')
cv=rep(cv,"int g_voc_lds_limit = 0;",'''}  // namespace fs2
extern "C" int fs2_dbg_cv_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fs2::g_cv_stamps), sizeof(unsigned long long) * 64 * 8 * 8);
}
namespace fs2 {
int g_voc_lds_limit = 0;''')
open('/tmp/fs2_stamp/vocoder_conv.hip','w').write(cv)
