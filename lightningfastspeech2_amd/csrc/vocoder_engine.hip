// HiFi-GAN generator engine behind the fs2_voc_* C ABI (include/fs2.h).
// Reference: litfass/third_party/hifigan/models.py:112-165 (Generator), :20-110 (ResBlock "1"),
// called by Synthesiser.__call__ (litfass/third_party/hifigan/__init__.py:37-43) from
// SpeechGenerator.generate_samples (litfass/synthesis/generator.py:163-170).
//
//   x = conv_pre(mel)                                              models.py:147
//   for each stage i:  x = ups[i](lrelu(x, 0.1))                   :149-150
//                      x = mean_j resblock[i, j](x)                :151-157
//       resblock: for (c1, c2): x = c2(lrelu(c1(lrelu(x)))) + x    :98-104
//   wav = tanh(conv_post(lrelu(x, 0.01)))                          :158-160  (F.leaky_relu default slope)
//
// Every conv is one launch of vocoder_conv_kernel; the LeakyReLU in front of a conv is applied
// while that conv stages its operand, the residual / the 1/3 mean over the resblocks in the epilogue
// of the conv that produces them.  ConvTranspose1d(k, stride s, padding p) with k - 2p = s:
//   y[q*s + f] = sum_m x[q - m] . Wt[:, :, m*s + f + p]   (0 <= m*s + f + p < k)
// is a plain conv of x to s*Cout channels (phase-major), whose (T, s*Cout) row-major output is the
// (T*s, Cout) tensor the next layer reads - no scatter, no zero insertion.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/fs2.h"
#include "fs2_common.h"
#include "fs2_kernels.h"

using namespace fs2;

namespace {

struct VocLayer {     // one conv in kernel form
    void* w = nullptr;
    float* b = nullptr;
    int cin = 0, cin_pad = 0, n = 0, taps = 1, dil = 1, pad = 0, wn = 1;
    int shift_from = 0;  // VocConvArgs::shift_from
};
struct HostT {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

}  // namespace

struct fs2_vocoder {
    fs2_voc_config cfg;
    int dt = FS2_BF16;
    size_t esz = 2;
    char err[512] = {0};
    bool finalized = false;
    std::map<std::string, std::vector<int64_t>> spec;
    std::map<std::string, HostT> host;
    std::vector<void*> allocs;
    VocLayer pre, post;
    std::vector<VocLayer> ups;
    std::vector<VocLayer> c1, c2;  // [(stage * n_kernels + j) * 3 + m]
    struct FusedRb { void* w = nullptr; float* b = nullptr; size_t conv_bytes = 0; };
    std::vector<FusedRb> rb;       // [stage * n_kernels + j]: six convs back to back (narrow stages)
    std::vector<int> chan, upf;    // channels / cumulative upsampling after stage i (index 0 = conv_pre)
    // workspace
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<void*> stage_out;  // x after conv_pre and after each stage
    int lastB = 0, lastT = 0;
};

namespace {

int vfail(fs2_vocoder* v, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(v->err, sizeof(v->err), fmt, ap);
    va_end(ap);
    return code;
}
#define VHIP(v, call)                                                                           \
    do {                                                                                        \
        hipError_t _s = (call);                                                                 \
        if (_s != hipSuccess) return vfail(v, FS2_ERR_HIP, "%s: %s", #call, hipGetErrorString(_s)); \
    } while (0)
#define VCHK(call)                     \
    do {                               \
        const int _r = (call);         \
        if (_r != FS2_OK) return _r;   \
    } while (0)

int vdev_alloc(fs2_vocoder* v, void** out, size_t bytes) {
    VHIP(v, hipMalloc(out, bytes ? bytes : 16));
    v->allocs.push_back(*out);
    return FS2_OK;
}

int next_pow2(int x) {
    int p = 32;
    while (p < x) p <<= 1;
    return p;
}

// W (n, cin, taps) fp32 host (conv form: out[t] = sum_tap x[t - pad + tap*dil] . W[:, :, tap]) ->
// fragment order [n-tile][step][wn][2][64] x 16 B as fp32 staging
void frag_order(int dt, const std::vector<float>& W, int n, int cin, int cin_pad, int taps, int wn_cols, bool post,
                std::vector<float>* stage) {
    const int KE = dt == FS2_BF16 ? 32 : 16, e16 = dt == FS2_BF16 ? 8 : 4;
    const int nkc = cin_pad / KE, nsteps4 = voc_steps_padded(taps, cin_pad, dt);
    const int ntiles = post ? 1 : n / (wn_cols * 32);
    const size_t nfrag = (size_t)ntiles * nsteps4 * wn_cols * 2 * 64;
    stage->assign(nfrag * e16, 0.f);
    for (int nt = 0; nt < ntiles; ++nt)
        for (int g = 0; g < taps * nkc; ++g) {
            const int tap = g / nkc, kc = g % nkc;
            for (int wn = 0; wn < wn_cols; ++wn)
                for (int ni = 0; ni < 2; ++ni)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int fr = lane & 15, fg = lane >> 4;
                        const int ch = nt * wn_cols * 32 + wn * 32 + (fr >> 2) * 8 + ni * 4 + (fr & 3);
                        if (ch >= n) continue;
                        float* dst = &(*stage)[((((size_t)nt * nsteps4 + g) * wn_cols + wn) * 2 + ni) * 64 * e16 + (size_t)lane * e16];
                        for (int e = 0; e < e16; ++e) {
                            const int c = kc * KE + fg * e16 + e;
                            if (c < cin) dst[e] = W[((size_t)ch * cin + c) * taps + tap];
                        }
                    }
        }
}

int upload_frags(fs2_vocoder* v, const std::vector<float>& stage, void** out) {
    if (v->dt == FS2_F32) {
        VCHK(vdev_alloc(v, out, stage.size() * 4));
        VHIP(v, hipMemcpy(*out, stage.data(), stage.size() * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<unsigned short> h(stage.size());
        for (size_t i = 0; i < stage.size(); ++i) h[i] = f32_to_bf16(stage[i]).v;
        VCHK(vdev_alloc(v, out, h.size() * 2));
        VHIP(v, hipMemcpy(*out, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    }
    return FS2_OK;
}

int pack_layer(fs2_vocoder* v, const std::vector<float>& W, const std::vector<float>& bias, int n, int cin, int taps,
               int dil, int pad, bool post, VocLayer* L) {
    L->cin = cin;
    L->cin_pad = next_pow2(cin);
    L->n = n;
    L->taps = taps;
    L->dil = dil;
    L->pad = pad;
    const int n32 = post ? 1 : n / 32;
    if (!post && n % 32) return vfail(v, FS2_ERR_SHAPE, "channel count %d is not a multiple of 32", n);
    L->wn = (n32 % 8 == 0) ? 8 : (n32 % 4 == 0) ? 4 : (n32 % 2 == 0) ? 2 : 1;
    std::vector<float> stage;
    frag_order(v->dt, W, n, cin, L->cin_pad, taps, L->wn, post, &stage);
    VCHK(upload_frags(v, stage, &L->w));
    std::vector<float> bp(post ? 32 : n, 0.f);
    for (size_t i = 0; i < bias.size() && i < bp.size(); ++i) bp[i] = bias[i];
    VCHK(vdev_alloc(v, (void**)&L->b, bp.size() * 4));
    VHIP(v, hipMemcpy(L->b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    return FS2_OK;
}

int conv_layer(fs2_vocoder* v, const std::string& name, int n, int cin, int k, int dil, bool post, VocLayer* L) {
    const HostT& w = v->host.at(name + ".weight");
    const HostT& b = v->host.at(name + ".bias");
    return pack_layer(v, w.data, b.data, n, cin, k, dil, (k - 1) / 2 * dil, post, L);
}

// ConvTranspose1d weight (cin, cout, k), stride s, padding p  ->  conv form (s*cout, cin, K)
int up_layer(fs2_vocoder* v, const std::string& name, int cin, int cout, int k, int s, VocLayer* L) {
    const HostT& w = v->host.at(name + ".weight");
    const HostT& b = v->host.at(name + ".bias");
    const int p = (k - s) / 2;
    if (k - 2 * p != s) return vfail(v, FS2_ERR_SHAPE, "%s: kernel %d / stride %d: kernel - 2*padding must equal the stride", name.c_str(), k, s);
    int m_lo = 1 << 30, m_hi = -(1 << 30);
    for (int f = 0; f < s; ++f)
        for (int m = -k; m <= k; ++m) {
            const int kk = m * s + f + p;
            if (kk >= 0 && kk < k) { m_lo = m < m_lo ? m : m_lo; m_hi = m > m_hi ? m : m_hi; }
        }
    const int K = m_hi - m_lo + 1, pad = m_hi;  // tap j reads x[q - pad + j]  <->  m = pad - j
    // A phase f only has the taps with 0 <= (pad - j)*s + f + p < k.  With k = 2s (every HiFi-GAN config) that is two of
    // the K = 3: j in {0, 1} for f < s/2, {1, 2} above - the kernel runs K - 1 taps and reads one row further on for
    // the channels from shift_from upwards (a wave column is 32 channels, cout a multiple of 32) instead of
    // multiplying a third of its steps by zeros.
    std::vector<int> jlo(s, K), jhi(s, -1);
    for (int f = 0; f < s; ++f)
        for (int j = 0; j < K; ++j) {
            const int kk = (pad - j) * s + f + p;
            if (kk >= 0 && kk < k) { jlo[f] = j < jlo[f] ? j : jlo[f]; jhi[f] = j > jhi[f] ? j : jhi[f]; }
        }
    int fshift = s;  // first phase whose window starts at tap 1
    bool two = K >= 2;
    for (int f = 0; f < s && two; ++f) {
        if (jlo[f] > 1 || jhi[f] - (jlo[f] >= 1 ? 1 : 0) > K - 2) two = false;
        if (jlo[f] >= 1 && fshift == s) fshift = f;
        if (jlo[f] < 1 && fshift != s) two = false;  // one switch point only
    }
    const int K2 = two ? K - 1 : K;
    std::vector<float> W((size_t)s * cout * cin * K2, 0.f), bias((size_t)s * cout);
    for (int f = 0; f < s; ++f)
        for (int co = 0; co < cout; ++co) {
            bias[(size_t)f * cout + co] = b.data[co];
            for (int j2 = 0; j2 < K2; ++j2) {
                const int j = j2 + (two && f >= fshift ? 1 : 0);
                const int kk = (pad - j) * s + f + p;
                if (kk < 0 || kk >= k) continue;
                for (int ci = 0; ci < cin; ++ci)
                    W[(((size_t)f * cout + co) * cin + ci) * K2 + j2] = w.data[((size_t)ci * cout + co) * k + kk];
            }
        }
    VCHK(pack_layer(v, W, bias, s * cout, cin, K2, 1, pad, false, L));
    L->shift_from = two && fshift < s ? fshift * cout : 0;
    return FS2_OK;
}

int run_conv(fs2_vocoder* v, hipStream_t st, const VocLayer& L, const void* x, void* out, const void* res,
             const int32_t* lengths, int len_scale, int B, int S, float in_slope, float scale, bool accumulate,
             bool in_fp32 = false, bool post = false, float out_slope = 1.f) {
    VocConvArgs a;
    a.tune = &op_tuning();  // the calling thread's A/B switches (fs2_op_set_vocoder_*)
    a.out_slope = out_slope;
    a.x = x; a.w = L.w; a.bias = L.b; a.res = res; a.out = out; a.lengths = lengths; a.len_scale = len_scale;
    a.B = B; a.S = S; a.cin = L.cin; a.cin_pad = L.cin_pad; a.n = L.n; a.taps = L.taps; a.dil = L.dil; a.pad = L.pad;
    a.wn = L.wn; a.in_slope = in_slope; a.scale = scale; a.accumulate = accumulate ? 1 : 0;
    a.in_fp32 = in_fp32 ? 1 : 0; a.post = post ? 1 : 0;
    a.shift_from = L.shift_from;
    const int r = launch_vocoder_conv(a, v->dt, st);
    if (r != FS2_OK) return vfail(v, r, "vocoder conv launch failed (cin=%d n=%d k=%d dil=%d S=%d)", L.cin, L.n, L.taps, L.dil, S);
    return FS2_OK;
}

}  // namespace

extern "C" {

int fs2_voc_create(const fs2_voc_config* cfg, fs2_vocoder** out) {
    if (!cfg || !out) return FS2_ERR_ARG;
    *out = nullptr;
    if (cfg->abi_version != FS2_ABI_VERSION) return FS2_ERR_ARG;
    if (cfg->n_stages < 1 || cfg->n_stages > FS2_VOC_MAX_STAGES || cfg->n_kernels < 1 || cfg->n_kernels > FS2_VOC_MAX_KERNELS)
        return FS2_ERR_SHAPE;
    if (cfg->n_mels < 4 || cfg->n_mels % 4 || (cfg->initial_channel >> cfg->n_stages) % 32) return FS2_ERR_SHAPE;
    if (cfg->dtype != FS2_F32 && cfg->dtype != FS2_BF16) return FS2_ERR_ARG;
    fs2_vocoder* v = new fs2_vocoder();
    v->cfg = *cfg;
    v->dt = cfg->dtype;
    v->esz = cfg->dtype == FS2_BF16 ? 2 : 4;
    const int C0 = cfg->initial_channel;
    v->spec["conv_pre.weight"] = {C0, cfg->n_mels, 7};
    v->spec["conv_pre.bias"] = {C0};
    v->chan.push_back(C0);
    v->upf.push_back(1);
    for (int i = 0; i < cfg->n_stages; ++i) {
        const int cin = C0 >> i, cout = C0 >> (i + 1);
        if (cfg->up_rates[i] < 1 || cfg->up_kernels[i] < cfg->up_rates[i]) { delete v; return FS2_ERR_SHAPE; }
        const std::string u = "ups." + std::to_string(i);
        v->spec[u + ".weight"] = {cin, cout, cfg->up_kernels[i]};
        v->spec[u + ".bias"] = {cout};
        v->chan.push_back(cout);
        v->upf.push_back(v->upf.back() * cfg->up_rates[i]);
        for (int j = 0; j < cfg->n_kernels; ++j) {
            const int k = cfg->rb_kernels[j];
            if (k < 1 || !(k & 1)) { delete v; return FS2_ERR_SHAPE; }
            const std::string r = "resblocks." + std::to_string(i * cfg->n_kernels + j);
            for (int m = 0; m < 3; ++m) {
                v->spec[r + ".convs1." + std::to_string(m) + ".weight"] = {cout, cout, k};
                v->spec[r + ".convs1." + std::to_string(m) + ".bias"] = {cout};
                v->spec[r + ".convs2." + std::to_string(m) + ".weight"] = {cout, cout, k};
                v->spec[r + ".convs2." + std::to_string(m) + ".bias"] = {cout};
            }
        }
    }
    v->spec["conv_post.weight"] = {1, v->chan.back(), 7};
    v->spec["conv_post.bias"] = {1};
    *out = v;
    return FS2_OK;
}

void fs2_voc_destroy(fs2_vocoder* v) {
    if (!v) return;
    for (void* p : v->allocs) (void)hipFree(p);
    if (v->ws) (void)hipFree(v->ws);
    delete v;
}

const char* fs2_voc_last_error(const fs2_vocoder* v) { return v ? v->err : "null vocoder"; }

int fs2_voc_load_weight(fs2_vocoder* v, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
    if (!v || !name || !data || !shape) return v ? vfail(v, FS2_ERR_ARG, "bad load_weight argument") : FS2_ERR_ARG;
    if (v->finalized) return vfail(v, FS2_ERR_STATE, "weights are frozen after fs2_voc_finalize");
    auto it = v->spec.find(name);
    if (it == v->spec.end()) return vfail(v, FS2_ERR_WEIGHT, "unknown weight '%s'", name);
    const auto& want = it->second;
    bool ok = (size_t)ndim == want.size();
    size_t n = 1;
    for (int i = 0; ok && i < ndim; ++i) { ok = shape[i] == want[i]; n *= (size_t)shape[i]; }
    if (!ok) return vfail(v, FS2_ERR_WEIGHT, "shape mismatch for '%s'", name);
    HostT t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + n);
    v->host[name] = std::move(t);
    return FS2_OK;
}

int fs2_voc_finalize(fs2_vocoder* v) {
    if (!v) return FS2_ERR_ARG;
    if (v->finalized) return FS2_OK;
    for (auto& kv : v->spec)
        if (!v->host.count(kv.first)) return vfail(v, FS2_ERR_WEIGHT, "missing weight '%s'", kv.first.c_str());
    const fs2_voc_config& c = v->cfg;
    VCHK(conv_layer(v, "conv_pre", c.initial_channel, c.n_mels, 7, 1, false, &v->pre));
    v->ups.resize(c.n_stages);
    v->c1.resize((size_t)c.n_stages * c.n_kernels * 3);
    v->c2.resize(v->c1.size());
    for (int i = 0; i < c.n_stages; ++i) {
        const int cin = v->chan[i], cout = v->chan[i + 1];
        VCHK(up_layer(v, "ups." + std::to_string(i), cin, cout, c.up_kernels[i], c.up_rates[i], &v->ups[i]));
        for (int j = 0; j < c.n_kernels; ++j) {
            const std::string r = "resblocks." + std::to_string(i * c.n_kernels + j);
            for (int m = 0; m < 3; ++m) {
                const size_t idx = ((size_t)i * c.n_kernels + j) * 3 + m;
                VCHK(conv_layer(v, r + ".convs1." + std::to_string(m), cout, cout, c.rb_kernels[j], c.rb_dilations[j][m], false, &v->c1[idx]));
                VCHK(conv_layer(v, r + ".convs2." + std::to_string(m), cout, cout, c.rb_kernels[j], 1, false, &v->c2[idx]));
            }
        }
    }
    VCHK(conv_layer(v, "conv_post", 1, v->chan.back(), 7, 1, true, &v->post));
    // narrow stages: the six convs of a resblock back to back for the LDS-resident kernel
    v->rb.resize((size_t)c.n_stages * c.n_kernels);
    for (int i = 0; i < c.n_stages; ++i) {
        const int C = v->chan[i + 1];
        if (C != 32 && C != 64 && C != 128) continue;
        for (int j = 0; j < c.n_kernels; ++j) {
            const std::string r = "resblocks." + std::to_string(i * c.n_kernels + j);
            std::vector<float> all, bias;
            for (int m = 0; m < 3; ++m)
                for (const char* grp : {".convs1.", ".convs2."}) {
                    const HostT& w = v->host.at(r + grp + std::to_string(m) + ".weight");
                    const HostT& b = v->host.at(r + grp + std::to_string(m) + ".bias");
                    std::vector<float> st;
                    frag_order(v->dt, w.data, C, C, C, c.rb_kernels[j], C / 32, false, &st);
                    all.insert(all.end(), st.begin(), st.end());
                    bias.insert(bias.end(), b.data.begin(), b.data.end());
                }
            fs2_vocoder::FusedRb& f = v->rb[(size_t)i * c.n_kernels + j];
            f.conv_bytes = all.size() / 6 * v->esz;
            VCHK(upload_frags(v, all, &f.w));
            VCHK(vdev_alloc(v, (void**)&f.b, bias.size() * 4));
            VHIP(v, hipMemcpy(f.b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
        }
    }
    v->host.clear();
    v->finalized = true;
    return FS2_OK;
}

int32_t fs2_voc_hop(const fs2_vocoder* v) { return v ? v->upf.back() : 0; }

int fs2_voc_synthesize(fs2_vocoder* v, const float* mel, const int32_t* lengths, int32_t B, int32_t T, float* wav,
                       void* stream) {
    if (!v || !mel || !wav || B <= 0 || T <= 0) return v ? vfail(v, FS2_ERR_ARG, "bad synthesize argument") : FS2_ERR_ARG;
    if (!v->finalized) return vfail(v, FS2_ERR_STATE, "fs2_voc_finalize not called");
    hipStream_t st = (hipStream_t)stream;
    const fs2_voc_config& c = v->cfg;
    const int ns = c.n_stages;
    // workspace: one output per stage (kept for the parity taps) + 4 scratch tensors of the largest stage
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t need = 0, big = 0;
    std::vector<size_t> sb(ns + 1);
    for (int i = 0; i <= ns; ++i) {
        sb[i] = al((size_t)B * T * v->upf[i] * v->chan[i] * v->esz);
        need += sb[i];
        if (i > 0 && sb[i] > big) big = sb[i];
    }
    need += 4 * big;
    if ((size_t)T * v->upf.back() > (size_t)0x7fffffff / 4) return vfail(v, FS2_ERR_SHAPE, "T too large");
    if (need > v->ws_bytes) {
        VHIP(v, hipStreamSynchronize(st));
        if (v->ws) (void)hipFree(v->ws);
        v->ws = nullptr;
        v->ws_bytes = 0;
        if (hipMalloc(&v->ws, need) != hipSuccess) return vfail(v, FS2_ERR_NOMEM, "workspace of %zu bytes", need);
        v->ws_bytes = need;
    }
    char* base = (char*)v->ws;
    v->stage_out.assign(ns + 1, nullptr);
    for (int i = 0; i <= ns; ++i) { v->stage_out[i] = base; base += sb[i]; }
    void* u = base;
    void* a = base + big;
    void* r1 = base + 2 * big;
    void* r2 = base + 3 * big;
    v->lastB = B;
    v->lastT = T;

    VCHK(run_conv(v, st, v->pre, mel, v->stage_out[0], nullptr, lengths, 1, B, T, 1.f, 1.f, false, /*in_fp32=*/true));
    const float inv = 1.0f / (float)c.n_kernels;
    for (int i = 0; i < ns; ++i) {
        // transposed conv, at the INPUT resolution, to up_rate * Cout phase-major channels
        const int S = T * v->upf[i + 1], sc = v->upf[i + 1];
        // when every resblock of this stage runs on LDS-resident tiles, the upsampled stream travels as lrelu(x):
        // all its consumers then fill their slabs by plain LDS-DMA (x_act, vocoder_resblock.hip)
        const int g_voc_fused_resblock = op_tuning().voc_fused_resblock;
        bool stage_act = g_voc_fused_resblock != 9;
        for (int j = 0; j < c.n_kernels && stage_act; ++j) {
            const fs2_vocoder::FusedRb& f = v->rb[(size_t)i * c.n_kernels + j];
            if (!f.w) { stage_act = false; break; }
            VocResblockArgs ra;
                ra.tune = &op_tuning();
            ra.x = u; ra.out = v->stage_out[i + 1]; ra.w = f.w; ra.bias = f.b; ra.lengths = lengths; ra.len_scale = sc;
            ra.B = B; ra.S = S; ra.C = v->chan[i + 1]; ra.taps = c.rb_kernels[j]; ra.wn = ra.C / 32; ra.npairs = 3;
            for (int m = 0; m < 3; ++m) ra.dil[m] = c.rb_dilations[j][m];
            ra.slope = 0.1f; ra.scale = 1.f; ra.accumulate = 0;
            if (voc_resblock_mi16(ra, v->dt)) continue;
            for (int m = 0; m < 3 && stage_act; ++m) {
                VocResblockArgs pa = ra;
                pa.npairs = 1;
                pa.dil[0] = c.rb_dilations[j][m];
                if (!voc_resblock_mi16(pa, v->dt)) stage_act = false;
            }
        }
        VCHK(run_conv(v, st, v->ups[i], v->stage_out[i], u, nullptr, lengths, v->upf[i], B, T * v->upf[i], 0.1f, 1.f, false, false,
                      false, stage_act ? 0.1f : 1.f));
        for (int j = 0; j < c.n_kernels; ++j) {
            const fs2_vocoder::FusedRb& f = v->rb[(size_t)i * c.n_kernels + j];
            if (f.w) {
                VocResblockArgs ra;
                ra.tune = &op_tuning();
                ra.x = u; ra.out = v->stage_out[i + 1]; ra.w = f.w; ra.bias = f.b; ra.lengths = lengths; ra.len_scale = sc;
                ra.B = B; ra.S = S; ra.C = v->chan[i + 1]; ra.taps = c.rb_kernels[j]; ra.wn = ra.C / 32; ra.npairs = 3;
                for (int m = 0; m < 3; ++m) ra.dil[m] = c.rb_dilations[j][m];
                ra.slope = 0.1f; ra.scale = inv; ra.accumulate = j > 0 ? 1 : 0;
                ra.x_act = stage_act ? 1 : 0;
                if (voc_resblock_mi16(ra, v->dt)) {  // the whole block in one launch
                    const int rr = launch_vocoder_resblock(ra, v->dt, st);
                    if (rr != FS2_OK) return vfail(v, rr, "fused resblock launch failed (C=%d k=%d)", ra.C, ra.taps);
                    continue;
                }
                // else pair by pair: a (c1, c2) pair on an LDS-resident tile where it fits and pays,
                // two plain convs otherwise
                const void* rin = u;
                bool resident[3];
                for (int m = 0; m < 3; ++m) {
                    VocResblockArgs pa = ra;
                    pa.npairs = 1;
                    pa.dil[0] = c.rb_dilations[j][m];
                    resident[m] = voc_resblock_mi16(pa, v->dt) != 0;
                }
                for (int m = 0; m < 3; ++m) {
                    void* o = m == 0 ? r1 : (m == 1 ? r2 : v->stage_out[i + 1]);
                    VocResblockArgs pa = ra;
                    pa.npairs = 1;
                    pa.dil[0] = c.rb_dilations[j][m];
                    pa.x = rin;
                    pa.out = o;
                    pa.w = (const char*)f.w + f.conv_bytes * 2 * m;
                    pa.bias = f.b + (size_t)2 * m * ra.C;
                    if (m < 2) { pa.scale = 1.f; pa.accumulate = 0; }
                    // between two resident pairs the residual stream travels as lrelu(x): the consumer's slab fill is
                    // then a plain LDS-DMA copy and it recovers x by the inverse map it applies anyway
                    pa.x_act = (m == 0 ? stage_act : (resident[m - 1] && resident[m] && g_voc_fused_resblock != 9)) ? 1 : 0;
                    pa.out_act = m < 2 && resident[m] && resident[m + 1] && g_voc_fused_resblock != 9;
                    if (resident[m]) {
                        const int rr = launch_vocoder_resblock(pa, v->dt, st);
                        if (rr != FS2_OK) return vfail(v, rr, "fused conv pair launch failed (C=%d k=%d)", ra.C, ra.taps);
                    } else {
                        const size_t idx = ((size_t)i * c.n_kernels + j) * 3 + m;
                        // c1 stores lrelu(out): c2 then stages it untouched (LDS-DMA fill in vocoder_conv.hip)
                        VCHK(run_conv(v, st, v->c1[idx], rin, a, nullptr, lengths, sc, B, S, 0.1f, 1.f, false, false, false, 0.1f));
                        VCHK(run_conv(v, st, v->c2[idx], a, o, rin, lengths, sc, B, S, 1.f, m < 2 ? 1.f : inv, m == 2 && j > 0));
                    }
                    rin = o;
                }
                continue;
            }
            const void* r = u;
            for (int m = 0; m < 3; ++m) {
                const size_t idx = ((size_t)i * c.n_kernels + j) * 3 + m;
                // c1 stores lrelu(out): c2 then stages it untouched (LDS-DMA fill in vocoder_conv.hip)
                VCHK(run_conv(v, st, v->c1[idx], r, a, nullptr, lengths, sc, B, S, 0.1f, 1.f, false, false, false, 0.1f));
                if (m < 2) {
                    void* o = m == 0 ? r1 : r2;
                    VCHK(run_conv(v, st, v->c2[idx], a, o, r, lengths, sc, B, S, 1.f, 1.f, false));
                    r = o;
                } else {  // last pair of the block: (xt + x) / n_kernels summed into the stage output
                    VCHK(run_conv(v, st, v->c2[idx], a, v->stage_out[i + 1], r, lengths, sc, B, S, 1.f, inv, j > 0));
                }
            }
        }
    }
    VCHK(run_conv(v, st, v->post, v->stage_out[ns], wav, nullptr, lengths, v->upf[ns], B, T * v->upf[ns], 0.01f, 1.f, false,
                  false, /*post=*/true));
    return FS2_OK;
}

int fs2_voc_debug_copy(fs2_vocoder* v, int32_t stage, float* dst, void* stream) {
    if (!v || !dst) return FS2_ERR_ARG;
    if (stage < 0 || stage > v->cfg.n_stages || v->stage_out.empty()) return vfail(v, FS2_ERR_STATE, "no such stage output");
    const size_t n = (size_t)v->lastB * v->lastT * v->upf[stage] * v->chan[stage];
    ConvertArgs a{v->stage_out[stage], dst, n};
    const int r = launch_convert(a, v->dt, FS2_F32, (hipStream_t)stream);
    if (r != FS2_OK) return vfail(v, r, "convert failed");
    return FS2_OK;
}

}  // extern "C"
