// Handle-level entry points (include/advshadow.h: advs_unet_*, advs_ddim_tables, advs_ddim_run): a host WITHOUT Python drives the
// eps-predictor (diff_model.py:163-267) and the DDIM loop (diff_model.py:416-474) through the same kernels, in the same order, as
// the Python plan (diff_model.py + engine.py of this package: emit_unet_forward / Builder, which this file restates in C++).
// Host code only: device memory from hipMalloc, one stream, the launch list replayed as one hipGraph.
#include "common.h"
#include <cmath>
#include <cstring>
#include <functional>
#include <list>
#include <map>
#include <string>
#include <vector>

namespace {

enum { K_STEM, K_RES, K_ATTN, K_DOWN, K_UP };
struct Layer { int kind; std::string p; int cin, cout; };
typedef std::vector<Layer> Stage;

struct Act { void* p = nullptr; int B = 0, H = 0, W = 0, C = 0; };      // NHWC activation of the compute dtype
struct StatsRef { float* p; int rbpi; };

// best-fit pool with explicit release (engine.py: Arena)
struct Arena {
    struct Blk { void* p; size_t n; };
    std::vector<Blk> all, free_list;
    int alloc(size_t nbytes, void** out) {
        nbytes = (nbytes + 255) / 256 * 256;
        int best = -1;
        for (size_t i = 0; i < free_list.size(); ++i) {
            const size_t n = free_list[i].n;
            if (n >= nbytes && n <= 2 * nbytes + 4096 && (best < 0 || n < free_list[best].n)) best = (int)i;
        }
        if (best >= 0) {
            *out = free_list[best].p;
            free_list.erase(free_list.begin() + best);
            return ADVS_OK;
        }
        void* p = nullptr;
        ADVS_HIP(hipMalloc(&p, nbytes));
        all.push_back({p, nbytes});
        *out = p;
        return ADVS_OK;
    }
    void release(void* p) {
        for (auto& b : all)
            if (b.p == p) { free_list.push_back(b); return; }
    }
    void destroy() {
        for (auto& b : all) (void)hipFree(b.p);
        all.clear();
        free_list.clear();
    }
};

struct Plan {
    std::vector<std::function<int(hipStream_t)>> ops;
    hipGraphExec_t graph = nullptr;
    int run_eager(hipStream_t s) {
        for (auto& f : ops) {
            const int rc = f(s);
            if (rc != ADVS_OK) return rc;
        }
        return ADVS_OK;
    }
    int capture(hipStream_t s) {
        ADVS_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = run_eager(s);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &g);
        if (rc != ADVS_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        const hipError_t e2 = hipGraphInstantiate(&graph, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
        return ADVS_OK;
    }
    int run(hipStream_t s) {
        if (graph) { ADVS_HIP(hipGraphLaunch(graph, s)); return ADVS_OK; }
        return run_eager(s);
    }
    void destroy() {
        if (graph) (void)hipGraphExecDestroy(graph);
        graph = nullptr;
        ops.clear();
    }
};

int esz(int dt) { return dt == ADVS_F32 ? 4 : 2; }
int slab(int dt) { return dt == ADVS_F32 ? 32 : 64; }

}  // namespace

struct advs_unet {
    advs_unet_config cfg;
    std::vector<Stage> down, up;
    Stage middle;
    std::vector<std::pair<std::string, long long>> params;        // state_dict keys in the reference's construction order
    std::map<std::string, std::vector<float>> host;               // what advs_unet_set_param received
    // ---- device weights (advs_unet_plan packs them once)
    std::map<std::string, void*> W;
    std::map<std::string, int> temb_off;
    int temb_total = 0;
    std::vector<void*> wbufs;
    bool packed = false;
    // ---- the plan of one forward for (batch, size)
    int B = 0, S = 0, uniform_t = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    std::list<advs_conv_args> cargs;                              // launch descriptors (stable addresses)
    std::map<void*, StatsRef> stats;
    Plan fwd, step;
    bool fwd_warm = false;
    float *x = nullptr, *eps = nullptr;
    long long* t = nullptr;
    void* gn_scratch = nullptr;
    // ---- DDIM loop
    float* coef = nullptr;
    long long* tseq = nullptr;
    int* counter = nullptr;
    int nsteps_cap = 0, step_clip = -1, step_nsteps = -1;
};

namespace {

// diff_model.py:190-237 flattened (the Python package's unet_layout)
void build_layout(advs_unet* u) {
    const advs_unet_config& c = u->cfg;
    auto in_attn = [&](int ds) {
        for (int i = 0; i < c.n_attention_resolutions; ++i)
            if (c.attention_resolutions[i] == ds) return true;
        return false;
    };
    const int mc = c.model_channels;
    u->down.push_back({{K_STEM, "down_blocks.0.0", c.in_channels, mc}});
    std::vector<int> skip_ch = {mc};
    int ch = mc, ds = 1;
    const int last = c.n_channel_mult - 1;
    for (int level = 0; level <= last; ++level) {
        const int mult = c.channel_mult[level];
        for (int r = 0; r < c.num_res_blocks; ++r) {
            const int idx = (int)u->down.size();
            Stage st = {{K_RES, "down_blocks." + std::to_string(idx) + ".0", ch, mult * mc}};
            ch = mult * mc;
            if (in_attn(ds)) st.push_back({K_ATTN, "down_blocks." + std::to_string(idx) + ".1", ch, ch});
            u->down.push_back(st);
            skip_ch.push_back(ch);
        }
        if (level != last) {
            u->down.push_back({{K_DOWN, "down_blocks." + std::to_string(u->down.size()) + ".0", ch, ch}});
            skip_ch.push_back(ch);
            ds *= 2;
        }
    }
    u->middle = {{K_RES, "middle_block.0", ch, ch}, {K_ATTN, "middle_block.1", ch, ch}, {K_RES, "middle_block.2", ch, ch}};
    for (int level = last; level >= 0; --level) {
        const int mult = c.channel_mult[level];
        for (int i = 0; i <= c.num_res_blocks; ++i) {
            const int idx = (int)u->up.size();
            const std::string b = "up_blocks." + std::to_string(idx) + ".";
            Stage st = {{K_RES, b + "0", ch + skip_ch.back(), mc * mult}};
            skip_ch.pop_back();
            ch = mc * mult;
            if (in_attn(ds)) st.push_back({K_ATTN, b + std::to_string(st.size()), ch, ch});
            if (level && i == c.num_res_blocks) {
                st.push_back({K_UP, b + std::to_string(st.size()), ch, ch});
                ds /= 2;
            }
            u->up.push_back(st);
        }
    }
}

template <typename F> void for_each_layer(advs_unet* u, F f) {
    for (auto& st : u->down) for (auto& l : st) f(l);
    for (auto& l : u->middle) f(l);
    for (auto& st : u->up) for (auto& l : st) f(l);
}

// the state_dict of diff_model.UNetModel (diff_model.py:163-243), names and element counts
void build_params(advs_unet* u) {
    const advs_unet_config& c = u->cfg;
    const long long mc = c.model_channels, ted = 4 * mc;
    auto add = [&](const std::string& n, long long numel) { u->params.push_back({n, numel}); };
    add("time_embed.0.weight", ted * mc); add("time_embed.0.bias", ted);
    add("time_embed.2.weight", ted * ted); add("time_embed.2.bias", ted);
    for_each_layer(u, [&](const Layer& l) {
        const long long ci = l.cin, co = l.cout;
        switch (l.kind) {
            case K_STEM: add(l.p + ".weight", co * ci * 9); add(l.p + ".bias", co); break;
            case K_RES:
                add(l.p + ".conv1.0.weight", ci); add(l.p + ".conv1.0.bias", ci);
                add(l.p + ".conv1.2.weight", co * ci * 9); add(l.p + ".conv1.2.bias", co);
                add(l.p + ".time_emb.1.weight", co * ted); add(l.p + ".time_emb.1.bias", co);
                add(l.p + ".conv2.0.weight", co); add(l.p + ".conv2.0.bias", co);
                add(l.p + ".conv2.3.weight", co * co * 9); add(l.p + ".conv2.3.bias", co);
                if (ci != co) { add(l.p + ".shortcut.weight", co * ci); add(l.p + ".shortcut.bias", co); }
                break;
            case K_ATTN:
                add(l.p + ".norm.weight", ci); add(l.p + ".norm.bias", ci);
                add(l.p + ".qkv.weight", 3 * ci * ci);
                add(l.p + ".proj.weight", ci * ci); add(l.p + ".proj.bias", ci);
                break;
            case K_DOWN: add(l.p + ".op.weight", ci * ci * 9); add(l.p + ".op.bias", ci); break;
            case K_UP: add(l.p + ".conv.weight", ci * ci * 9); add(l.p + ".conv.bias", ci); break;
        }
    });
    add("out.0.weight", mc); add("out.0.bias", mc);
    add("out.2.weight", (long long)c.out_channels * mc * 9); add("out.2.bias", c.out_channels);
}

int dev_copy(advs_unet* u, const float* src, size_t n, void** out) {
    void* p = nullptr;
    ADVS_HIP(hipMalloc(&p, n * sizeof(float)));
    u->wbufs.push_back(p);
    ADVS_HIP(hipMemcpy(p, src, n * sizeof(float), hipMemcpyHostToDevice));
    *out = p;
    return ADVS_OK;
}
int dev_f32(advs_unet* u, const std::string& key, const std::string& name) {
    const std::vector<float>& v = u->host.at(name);
    void* p = nullptr;
    const int rc = dev_copy(u, v.data(), v.size(), &p);
    if (rc == ADVS_OK) u->W[key] = p;
    return rc;
}
// OIHW f32 (host) -> [O][R][S][I] in the compute dtype (advs_pack_conv_weight), optionally into a caller-provided destination
int pack_conv(advs_unet* u, const float* w_host, int O, int I, int R, int S, void** out, void* dst = nullptr) {
    void* tmp = nullptr;
    const size_t n = (size_t)O * I * R * S;
    ADVS_HIP(hipMalloc(&tmp, n * sizeof(float)));
    hipError_t e = hipMemcpy(tmp, w_host, n * sizeof(float), hipMemcpyHostToDevice);
    void* p = dst;
    if (e == hipSuccess && !p) {
        e = hipMalloc(&p, n * esz(u->cfg.dtype));
        if (e == hipSuccess) u->wbufs.push_back(p);
    }
    int rc = ADVS_OK;
    if (e == hipSuccess) rc = advs_pack_conv_weight((const float*)tmp, p, O, I, R, S, u->cfg.dtype, u->stream);
    if (e == hipSuccess && rc == ADVS_OK) e = hipStreamSynchronize(u->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "pack_conv: %s", hipGetErrorString(e));
    if (rc == ADVS_OK && out) *out = p;
    return rc;
}
int pack_named(advs_unet* u, const std::string& key, const std::string& name, int O, int I, int R) {
    void* p = nullptr;
    const int rc = pack_conv(u, u->host.at(name).data(), O, I, R, R, &p);
    if (rc == ADVS_OK) u->W[key] = p;
    return rc;
}

// OIHW 3x3 weight of a conv behind a nearest x2 upsample -> [4][O][2][2][I]: for output parity (a, b) the taps that read the same
// low-res pixel are summed in f32 before rounding -- rows first, then columns, the order of engine.pack_subpixel_upsample_weight
int pack_subpixel(advs_unet* u, const std::string& key, const std::string& name, int O, int I) {
    const std::vector<float>& w = u->host.at(name);
    const size_t per = (size_t)O * I * 4;
    void* dst = nullptr;
    ADVS_HIP(hipMalloc(&dst, 4 * per * esz(u->cfg.dtype)));
    u->wbufs.push_back(dst);
    std::vector<float> c(per);
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            for (size_t oi = 0; oi < (size_t)O * I; ++oi) {
                const float* k = &w[oi * 9];
                float r[2][3];
                for (int s = 0; s < 3; ++s) {
                    if (a == 0) { r[0][s] = k[s]; r[1][s] = k[3 + s] + k[6 + s]; }
                    else { r[0][s] = k[s] + k[3 + s]; r[1][s] = k[6 + s]; }
                }
                for (int y = 0; y < 2; ++y) {
                    if (b == 0) { c[oi * 4 + y * 2] = r[y][0]; c[oi * 4 + y * 2 + 1] = r[y][1] + r[y][2]; }
                    else { c[oi * 4 + y * 2] = r[y][0] + r[y][1]; c[oi * 4 + y * 2 + 1] = r[y][2]; }
                }
            }
            const int rc = pack_conv(u, c.data(), O, I, 2, 2, nullptr, (char*)dst + (size_t)(2 * a + b) * per * esz(u->cfg.dtype));
            if (rc != ADVS_OK) return rc;
        }
    u->W[key] = dst;
    return ADVS_OK;
}

// UNetModel.packed_weights (this package's diff_model.py)
int pack_weights(advs_unet* u) {
    if (u->packed) return ADVS_OK;
    for (auto& pr : u->params)
        ADVS_REQUIRE(u->host.count(pr.first), "advs_unet_plan: parameter %s was never set (advs_unet_set_param)", pr.first.c_str());
    const int dt = u->cfg.dtype, sl = slab(dt);
    std::vector<float> temb_w, temb_b;
    int off = 0, rc = ADVS_OK;
    std::vector<const Layer*> layers;
    for_each_layer(u, [&](const Layer& l) { layers.push_back(&l); });
    for (const Layer* lp : layers) {
        const Layer& l = *lp;
        const std::string& p = l.p;
        if (l.kind == K_STEM) {
            if ((rc = dev_f32(u, p + ".w", p + ".weight")) || (rc = dev_f32(u, p + ".b", p + ".bias"))) return rc;
        } else if (l.kind == K_RES) {
            ADVS_REQUIRE(l.cin % sl == 0 && l.cout % sl == 0, "%s: channels (%d->%d) must be multiples of %d for this dtype", p.c_str(), l.cin, l.cout, sl);
            for (const char* n : {".conv1.0", ".conv2.0"})
                if ((rc = dev_f32(u, p + n + ".g", p + n + ".weight")) || (rc = dev_f32(u, p + n + ".b", p + n + ".bias"))) return rc;
            if ((rc = pack_named(u, p + ".conv1.2.w", p + ".conv1.2.weight", l.cout, l.cin, 3))) return rc;
            if ((rc = dev_f32(u, p + ".conv1.2.b", p + ".conv1.2.bias"))) return rc;
            if (l.cin == l.cout) {
                if ((rc = pack_named(u, p + ".conv2.3.w", p + ".conv2.3.weight", l.cout, l.cout, 3))) return rc;
                if ((rc = dev_f32(u, p + ".conv2.3.b", p + ".conv2.3.bias"))) return rc;
            } else {
                // shortcut fused into conv2: K = [9*cout | cin] per output channel, one bias (diff_model.py:102-103)
                void *w3 = nullptr, *w1 = nullptr, *wf = nullptr;
                if ((rc = pack_conv(u, u->host.at(p + ".conv2.3.weight").data(), l.cout, l.cout, 3, 3, &w3))) return rc;
                if ((rc = pack_conv(u, u->host.at(p + ".shortcut.weight").data(), l.cout, l.cin, 1, 1, &w1))) return rc;
                const size_t k3 = (size_t)9 * l.cout * esz(dt), k1 = (size_t)l.cin * esz(dt);
                ADVS_HIP(hipMalloc(&wf, (k3 + k1) * l.cout));
                u->wbufs.push_back(wf);
                ADVS_HIP(hipMemcpy2D(wf, k3 + k1, w3, k3, k3, l.cout, hipMemcpyDeviceToDevice));
                ADVS_HIP(hipMemcpy2D((char*)wf + k3, k3 + k1, w1, k1, k1, l.cout, hipMemcpyDeviceToDevice));
                u->W[p + ".conv2.3.w"] = wf;
                std::vector<float> bsum(u->host.at(p + ".conv2.3.bias"));
                const std::vector<float>& sb = u->host.at(p + ".shortcut.bias");
                for (int i = 0; i < l.cout; ++i) bsum[i] = bsum[i] + sb[i];
                void* bp = nullptr;
                if ((rc = dev_copy(u, bsum.data(), bsum.size(), &bp))) return rc;
                u->W[p + ".conv2.3.b"] = bp;
            }
            const std::vector<float>& tw = u->host.at(p + ".time_emb.1.weight");
            const std::vector<float>& tb = u->host.at(p + ".time_emb.1.bias");
            temb_w.insert(temb_w.end(), tw.begin(), tw.end());
            temb_b.insert(temb_b.end(), tb.begin(), tb.end());
            u->temb_off[p] = off;
            off += l.cout;
        } else if (l.kind == K_ATTN) {
            if ((rc = dev_f32(u, p + ".norm.g", p + ".norm.weight")) || (rc = dev_f32(u, p + ".norm.b", p + ".norm.bias"))) return rc;
            if ((rc = pack_named(u, p + ".qkv.w", p + ".qkv.weight", 3 * l.cin, l.cin, 1))) return rc;
            if ((rc = pack_named(u, p + ".proj.w", p + ".proj.weight", l.cin, l.cin, 1))) return rc;
            if ((rc = dev_f32(u, p + ".proj.b", p + ".proj.bias"))) return rc;
        } else if (l.kind == K_DOWN) {
            if ((rc = pack_named(u, p + ".op.w", p + ".op.weight", l.cin, l.cin, 3)) || (rc = dev_f32(u, p + ".op.b", p + ".op.bias"))) return rc;
        } else if (l.kind == K_UP) {
            if ((rc = pack_named(u, p + ".conv.w", p + ".conv.weight", l.cin, l.cin, 3))) return rc;
            if ((rc = pack_subpixel(u, p + ".conv.w4", p + ".conv.weight", l.cin, l.cin))) return rc;
            if ((rc = dev_f32(u, p + ".conv.b", p + ".conv.bias"))) return rc;
        }
    }
    void* q = nullptr;
    if ((rc = dev_copy(u, temb_w.data(), temb_w.size(), &q))) return rc;
    u->W["temb_w"] = q;
    if ((rc = dev_copy(u, temb_b.data(), temb_b.size(), &q))) return rc;
    u->W["temb_b"] = q;
    u->temb_total = off;
    for (const char* k : {"time_embed.0", "time_embed.2", "out.2"})
        if ((rc = dev_f32(u, std::string(k) + ".w", std::string(k) + ".weight")) || (rc = dev_f32(u, std::string(k) + ".b", std::string(k) + ".bias"))) return rc;
    if ((rc = dev_f32(u, "out.0.g", "out.0.weight")) || (rc = dev_f32(u, "out.0.b", "out.0.bias"))) return rc;
    // frequency table of the sinusoidal embedding (diff_model.py:26-28): exp(-ln(10000) * i / half) in f32.  A caller that wants the
    // bits of another framework's table (torch's vectorised expf may differ from this one in the last place) sets "freqs".
    const int half = u->cfg.model_channels / 2;
    std::vector<float> fr(half);
    if (u->host.count("freqs")) fr = u->host.at("freqs");
    else
        for (int i = 0; i < half; ++i) fr[i] = expf((float)(-log(10000.0)) * (float)i / (float)half);
    if ((rc = dev_copy(u, fr.data(), fr.size(), &q))) return rc;
    u->W["freqs"] = q;
    u->packed = true;
    return ADVS_OK;
}

// ---- the plan builder (engine.py: Builder) -------------------------------------------------------------------------------------
struct Bld {
    advs_unet* u;
    int rc = ADVS_OK;
    int dt() const { return u->cfg.dtype; }
    void* alloc(size_t bytes) {
        void* p = nullptr;
        if (rc == ADVS_OK) rc = u->arena.alloc(bytes, &p);
        return p;
    }
    Act buf(int B, int H, int W, int C) {
        Act a;
        a.p = alloc((size_t)B * H * W * C * esz(dt()));
        a.B = B; a.H = H; a.W = W; a.C = C;
        return a;
    }
    void free(const Act& a) { free_ptr(a.p); }
    void free_ptr(void* p) {
        if (!p) return;
        u->arena.release(p);
        auto it = u->stats.find(p);
        if (it != u->stats.end()) {
            u->arena.release(it->second.p);
            u->stats.erase(it);
        }
    }
    void* Wt(const std::string& k) { return u->W.at(k); }

    struct ConvOpt {
        const Act* x2 = nullptr; const float* bias = nullptr; const float* temb = nullptr; int temb_stride = 0; const Act* residual = nullptr;
        int ksize = 3, stride = 1, pad = 1, upsample = 0; bool want_stats = false; const Act* e1 = nullptr; const Act* e2 = nullptr;
        const float* norm = nullptr;
    };
    Act conv(const Act& x1, const void* w, int cout, const ConvOpt& o) {
        const int HL = o.upsample ? 2 * x1.H : x1.H, WL = o.upsample ? 2 * x1.W : x1.W;
        const int Ho = (HL + 2 * o.pad - o.ksize) / o.stride + 1, Wo = (WL + 2 * o.pad - o.ksize) / o.stride + 1;
        Act y = buf(x1.B, Ho, Wo, cout);
        u->cargs.emplace_back();
        advs_conv_args& a = u->cargs.back();
        memset(&a, 0, sizeof(a));
        a.x1 = x1.p; a.x2 = o.x2 ? o.x2->p : nullptr; a.w = w; a.bias = o.bias; a.temb = o.temb;
        a.residual = o.residual ? o.residual->p : nullptr; a.y = y.p;
        a.b = x1.B; a.h = x1.H; a.w_ = x1.W; a.c1 = x1.C; a.c2 = o.x2 ? o.x2->C : 0; a.cout = cout;
        a.ksize = o.ksize; a.stride = o.stride; a.pad = o.pad; a.upsample = o.upsample;
        a.act = ADVS_ACT_NONE; a.dtype = dt(); a.temb_stride = o.temb_stride;
        a.e1 = o.e1 ? o.e1->p : nullptr; a.e2 = o.e2 ? o.e2->p : nullptr; a.ce1 = o.e1 ? o.e1->C : 0; a.ce2 = o.e2 ? o.e2->C : 0;
        a.norm = o.norm;
        if (o.want_stats) {
            a.tile = advs_conv_resolve_tile(&a);
            const int rows = advs_conv_tile_rows(a.tile);
            if (rows > 0 && (Ho * Wo) % rows == 0) {
                float* st = (float*)alloc((size_t)x1.B * (Ho * Wo / rows) * cout * 2 * sizeof(float));
                u->stats[y.p] = {st, Ho * Wo / rows};
                a.stats = st; a.stats_rows = rows;
            }
        }
        const advs_conv_args* ap = &a;
        u->fwd.ops.push_back([ap](hipStream_t s) { return advs_conv2d(ap, s); });
        return y;
    }
    Act groupnorm(const Act& x, const float* g, const float* b, int act, const Act* x2 = nullptr) {
        const int C2 = x2 ? x2->C : 0;
        Act y = buf(x.B, x.H, x.W, x.C + C2);
        auto s1 = u->stats.find(x.p);
        auto s2 = x2 ? u->stats.find(x2->p) : u->stats.end();
        const void *xp = x.p, *x2p = x2 ? x2->p : nullptr;
        void *yp = y.p, *scr = u->gn_scratch;
        const int B = x.B, hw = x.H * x.W, C1 = x.C, d = dt();
        if (s1 != u->stats.end() && (!x2 || s2 != u->stats.end())) {
            const StatsRef a1 = s1->second, a2 = x2 ? s2->second : StatsRef{nullptr, 0};
            u->fwd.ops.push_back([=](hipStream_t s) {
                return advs_groupnorm_stats(xp, x2p, a1.p, a1.rbpi, a2.p, a2.rbpi, g, b, nullptr, nullptr, 0, yp, scr, B, hw, C1, C2, 32, act, d, s);
            });
        } else {
            u->fwd.ops.push_back([=](hipStream_t s) { return advs_groupnorm(xp, x2p, g, b, nullptr, nullptr, 0, yp, scr, B, hw, C1, C2, 32, act, d, s); });
        }
        return y;
    }
    // engine.Builder.can_fuse_norm: GroupNorm + SiLU inside the conv that reads it (advs_conv_args.norm)
    bool can_fuse_norm(const Act& x, const Act* x2, int cout) {
        const int C2 = x2 ? x2->C : 0;
        if (dt() == ADVS_F32 || cout > 128 || x.C + C2 > 384 || x.C % 64 || C2 % 64) return false;
        if (x.H % 16 || x.W % 16 || (long long)x.H * x.W < 128 * 128) return false;
        return u->stats.count(x.p) && (!x2 || u->stats.count(x2->p));
    }
    float* groupnorm_affine(const Act& x, const float* g, const float* b, const Act* x2) {
        const int C2 = x2 ? x2->C : 0;
        const StatsRef a1 = u->stats.at(x.p), a2 = x2 ? u->stats.at(x2->p) : StatsRef{nullptr, 0};
        float* table = (float*)alloc((size_t)x.B * (x.C + C2) * 2 * sizeof(float));
        void* scr = u->gn_scratch;
        const int B = x.B, hw = x.H * x.W, C1 = x.C;
        u->fwd.ops.push_back([=](hipStream_t s) { return advs_groupnorm_affine_stats(a1.p, a1.rbpi, a2.p, a2.rbpi, g, b, scr, table, B, hw, C1, C2, 32, s); });
        return table;
    }
    float* linear(const float* x, int Bn, int K, const float* w, const float* bias, int N, int act_in, int act_out) {
        float* y = (float*)alloc((size_t)Bn * N * sizeof(float));
        u->fwd.ops.push_back([=](hipStream_t s) { return advs_linear_f32(x, w, bias, y, Bn, K, N, act_in, act_out, s); });
        return y;
    }
};

// emit_unet_forward (this package's diff_model.py), line for line
int emit_forward(advs_unet* u) {
    Bld b{u};
    const advs_unet_config& c = u->cfg;
    const int B = u->B, mc = c.model_channels, ted = 4 * mc, half = mc / 2, heads = c.num_heads;
    const int rows = u->uniform_t ? 1 : B;
    float* e0 = (float*)b.alloc((size_t)rows * 2 * half * sizeof(float));
    {
        const long long* t = u->t;
        const float* fr = (const float*)b.Wt("freqs");
        u->fwd.ops.push_back([=](hipStream_t s) { return advs_timestep_embedding((const int64_t*)t, fr, half, 1, nullptr, nullptr, e0, rows, s); });
    }
    float* e1 = b.linear(e0, rows, mc, (const float*)b.Wt("time_embed.0.w"), (const float*)b.Wt("time_embed.0.b"), ted, ADVS_ACT_NONE, ADVS_ACT_SILU);
    float* emb = b.linear(e1, rows, ted, (const float*)b.Wt("time_embed.2.w"), (const float*)b.Wt("time_embed.2.b"), ted, ADVS_ACT_NONE, ADVS_ACT_NONE);
    float* temb = b.linear(emb, rows, ted, (const float*)b.Wt("temb_w"), (const float*)b.Wt("temb_b"), u->temb_total, ADVS_ACT_SILU, ADVS_ACT_NONE);
    const int tstride = u->uniform_t ? -1 : u->temb_total;

    auto norm_silu_conv = [&](const Act& x1, const Act* x2, const std::string& gk, const std::string& wk, int cout, Bld::ConvOpt o) {
        const float *g = (const float*)b.Wt(gk + ".g"), *be = (const float*)b.Wt(gk + ".b");
        o.bias = (const float*)b.Wt(wk + ".b");
        o.want_stats = true;
        if (b.can_fuse_norm(x1, x2, cout)) {
            float* tab = b.groupnorm_affine(x1, g, be, x2);
            o.x2 = x2;
            o.norm = tab;
            Act y = b.conv(x1, b.Wt(wk + ".w"), cout, o);
            b.free_ptr(tab);
            return y;
        }
        Act a = b.groupnorm(x1, g, be, ADVS_ACT_SILU, x2);
        Act y = b.conv(a, b.Wt(wk + ".w"), cout, o);
        b.free(a);
        return y;
    };
    auto res_block = [&](const Layer& l, const Act& x1, const Act* x2) {
        Bld::ConvOpt o1;
        o1.temb = temb + u->temb_off.at(l.p);
        o1.temb_stride = tstride;
        Act h1 = norm_silu_conv(x1, x2, l.p + ".conv1.0", l.p + ".conv1.2", l.cout, o1);
        Bld::ConvOpt o2;
        if (l.cin != l.cout) { o2.e1 = &x1; o2.e2 = x2; }        // conv2(h) + shortcut(cat[x1, x2]) as one implicit GEMM
        else o2.residual = &x1;
        Act y = norm_silu_conv(h1, nullptr, l.p + ".conv2.0", l.p + ".conv2.3", l.cout, o2);
        b.free(h1);
        return y;
    };
    auto attn_block = [&](const Layer& l, const Act& x) {
        const int ch = l.cin, d = ch / heads;
        Act n = b.groupnorm(x, (const float*)b.Wt(l.p + ".norm.g"), (const float*)b.Wt(l.p + ".norm.b"), ADVS_ACT_NONE);
        Bld::ConvOpt oq;
        oq.ksize = 1; oq.pad = 0;
        Act qkv = b.conv(n, b.Wt(l.p + ".qkv.w"), 3 * ch, oq);
        b.free(n);
        Act o = b.buf(x.B, x.H, x.W, ch);
        {
            const void* qp = qkv.p;
            void* op = o.p;
            const int N = x.H * x.W, Bq = x.B, ld = 3 * ch, dtv = b.dt();
            // per head the 3d output channels are [q | k | v] (reshape + chunk, diff_model.py:120)
            u->fwd.ops.push_back([=](hipStream_t s) { return advs_attention_masked(qp, op, Bq, N, N, heads, d, ld, 0, d, 2 * d, 3 * d, dtv, s); });
        }
        b.free(qkv);
        Bld::ConvOpt op2;
        op2.ksize = 1; op2.pad = 0; op2.bias = (const float*)b.Wt(l.p + ".proj.b"); op2.residual = &x; op2.want_stats = true;
        Act y = b.conv(o, b.Wt(l.p + ".proj.w"), ch, op2);
        b.free(o);
        return y;
    };

    std::vector<Act> hs;
    auto run_stage = [&](const Stage& st, Act h, const Act* skip) {
        for (const Layer& l : st) {
            Act nw;
            if (l.kind == K_STEM) {
                nw = b.buf(B, u->S, u->S, l.cout);
                float* stp = nullptr;
                const int rws = advs_conv_first_stats_rows(l.cin, u->S, u->S, l.cout, b.dt());
                if (rws > 0) {
                    stp = (float*)b.alloc((size_t)B * (u->S * u->S / rws) * l.cout * 2 * sizeof(float));
                    u->stats[nw.p] = {stp, u->S * u->S / rws};
                }
                const float *xp = u->x, *w = (const float*)b.Wt(l.p + ".w"), *bi = (const float*)b.Wt(l.p + ".b");
                void* yp = nw.p;
                const int cin = l.cin, S = u->S, co = l.cout, dtv = b.dt();
                u->fwd.ops.push_back([=](hipStream_t s) { return advs_conv3x3_first_stats(xp, w, bi, yp, stp, B, cin, S, S, co, dtv, s); });
            } else if (l.kind == K_RES) {
                nw = res_block(l, h, skip);
                skip = nullptr;
            } else if (l.kind == K_ATTN) {
                nw = attn_block(l, h);
            } else if (l.kind == K_DOWN) {
                Bld::ConvOpt o;
                o.bias = (const float*)b.Wt(l.p + ".op.b"); o.stride = 2; o.want_stats = true;
                nw = b.conv(h, b.Wt(l.p + ".op.w"), l.cout, o);
            } else {
                // Upsample (diff_model.py:129-140): nearest x2 + 3x3, computed on the low-res grid where the shape allows
                Bld::ConvOpt o;
                o.bias = (const float*)b.Wt(l.p + ".conv.b"); o.want_stats = true;
                const bool sub = h.H % 16 == 0 && h.W % 16 == 0;
                o.upsample = sub ? ADVS_UPSAMPLE_SUBPIXEL : 1;
                nw = b.conv(h, b.Wt(l.p + (sub ? ".conv.w4" : ".conv.w")), l.cout, o);
            }
            bool kept = false;
            for (auto& s : hs) kept = kept || s.p == h.p;
            if (h.p && !kept) b.free(h);
            h = nw;
        }
        return h;
    };
    Act h;
    for (auto& st : u->down) {
        h = run_stage(st, h, nullptr);
        hs.push_back(h);
    }
    h = run_stage(u->middle, h, nullptr);
    for (auto& st : u->up) {
        Act skip = hs.back();
        hs.pop_back();
        Act h2 = run_stage(st, h, &skip);        // first layer is the res block reading cat([h, skip])
        b.free(skip);
        h = h2;
    }
    Act a = b.groupnorm(h, (const float*)b.Wt("out.0.g"), (const float*)b.Wt("out.0.b"), ADVS_ACT_SILU);
    b.free(h);
    {
        const void* ap = a.p;
        const float *w = (const float*)b.Wt("out.2.w"), *bi = (const float*)b.Wt("out.2.b");
        float* out = u->eps;
        const int S = u->S, co = c.out_channels, dtv = b.dt();
        u->fwd.ops.push_back([=](hipStream_t s) { return advs_conv_last(ap, w, bi, out, B, mc, S, S, co, 3, dtv, s); });
    }
    b.free(a);
    return b.rc;
}

void drop_plan(advs_unet* u) {
    u->fwd.destroy();
    u->step.destroy();
    u->arena.destroy();
    u->cargs.clear();
    u->stats.clear();
    for (void* p : {(void*)u->x, (void*)u->eps, (void*)u->t, u->gn_scratch, (void*)u->coef, (void*)u->tseq, (void*)u->counter})
        if (p) (void)hipFree(p);
    u->x = u->eps = nullptr; u->t = nullptr; u->gn_scratch = nullptr; u->coef = nullptr; u->tseq = nullptr; u->counter = nullptr;
    u->B = u->S = 0; u->fwd_warm = false; u->nsteps_cap = 0; u->step_clip = -1; u->step_nsteps = -1;
}

}  // namespace

extern "C" int advs_unet_create(const advs_unet_config* cfg, advs_unet** out) {
    ADVS_REQUIRE(cfg && out, "advs_unet_create: null pointer");
    ADVS_REQUIRE(cfg->in_channels > 0 && cfg->in_channels <= 4 && cfg->out_channels > 0 && cfg->out_channels <= 4, "advs_unet_create: 1..4 image channels");
    ADVS_REQUIRE(cfg->model_channels > 0 && cfg->model_channels % 32 == 0 && cfg->num_res_blocks > 0 && cfg->num_heads > 0, "advs_unet_create: bad widths");
    ADVS_REQUIRE(cfg->n_channel_mult >= 1 && cfg->n_channel_mult <= 8 && cfg->n_attention_resolutions >= 0 && cfg->n_attention_resolutions <= 8,
                 "advs_unet_create: 1..8 levels, at most 8 attention resolutions");
    ADVS_REQUIRE(cfg->dtype == ADVS_F32 || cfg->dtype == ADVS_BF16 || cfg->dtype == ADVS_F16, "advs_unet_create: unknown dtype %d", cfg->dtype);
    advs_unet* u = new advs_unet();
    u->cfg = *cfg;
    build_layout(u);
    build_params(u);
    *out = u;
    return ADVS_OK;
}

extern "C" int advs_unet_param_count(const advs_unet* u) { return u ? (int)u->params.size() : 0; }

extern "C" int advs_unet_param_name(const advs_unet* u, int i, char* name, int name_len, long long* numel) {
    ADVS_REQUIRE(u && i >= 0 && i < (int)u->params.size() && name && name_len > 0, "advs_unet_param_name: bad index");
    ADVS_REQUIRE((int)u->params[i].first.size() < name_len, "advs_unet_param_name: %d bytes are too few for %s", name_len, u->params[i].first.c_str());
    snprintf(name, (size_t)name_len, "%s", u->params[i].first.c_str());
    if (numel) *numel = u->params[i].second;
    return ADVS_OK;
}

extern "C" int advs_unet_set_param(advs_unet* u, const char* name, const float* host_data, long long numel) {
    ADVS_REQUIRE(u && name && host_data, "advs_unet_set_param: null pointer");
    long long want = -1;
    if (!strcmp(name, "freqs")) want = u->cfg.model_channels / 2;
    for (auto& pr : u->params)
        if (pr.first == name) want = pr.second;
    ADVS_REQUIRE(want >= 0, "advs_unet_set_param: %s is not a parameter of this network", name);
    ADVS_REQUIRE(numel == want, "advs_unet_set_param: %s has %lld elements, got %lld", name, want, numel);
    u->host[name].assign(host_data, host_data + numel);
    if (u->packed) {                                  // new weights: the packed copies and every plan that points into them go
        drop_plan(u);
        for (void* p : u->wbufs) (void)hipFree(p);
        u->wbufs.clear(); u->W.clear(); u->temb_off.clear(); u->packed = false;
    }
    return ADVS_OK;
}

extern "C" int advs_unet_plan(advs_unet* u, int batch, int size, int uniform_t, void* stream) {
    ADVS_REQUIRE(u && batch > 0 && size > 0 && stream, "advs_unet_plan: bad arguments (an explicit stream is needed: the plan is captured into a hipGraph)");
    int rc = advs_init();
    if (rc != ADVS_OK) return rc;
    if (u->B) drop_plan(u);
    u->stream = (hipStream_t)stream;
    if ((rc = pack_weights(u))) return rc;
    u->B = batch; u->S = size; u->uniform_t = uniform_t ? 1 : 0;
    const size_t img = (size_t)size * size;
    ADVS_HIP(hipMalloc((void**)&u->x, batch * u->cfg.in_channels * img * sizeof(float)));
    ADVS_HIP(hipMalloc((void**)&u->eps, batch * u->cfg.out_channels * img * sizeof(float)));
    ADVS_HIP(hipMalloc((void**)&u->t, batch * sizeof(long long)));
    ADVS_HIP(hipMalloc(&u->gn_scratch, advs_groupnorm_scratch_bytes(batch, 64)));
    ADVS_HIP(hipMemset(u->t, 0, batch * sizeof(long long)));
    ADVS_HIP(hipMemset(u->x, 0, batch * u->cfg.in_channels * img * sizeof(float)));
    if ((rc = emit_forward(u))) { drop_plan(u); return rc; }
    // first replay eager (module load, attribute opt-ins), then captured
    if ((rc = u->fwd.run_eager(u->stream))) { drop_plan(u); return rc; }
    ADVS_HIP(hipStreamSynchronize(u->stream));
    if ((rc = u->fwd.capture(u->stream))) { drop_plan(u); return rc; }
    u->fwd_warm = true;
    return ADVS_OK;
}

extern "C" int advs_unet_forward(advs_unet* u, const float* x_nchw, const int64_t* t, float* eps_nchw) {
    ADVS_REQUIRE(u && x_nchw && t && eps_nchw, "advs_unet_forward: null pointer");
    ADVS_REQUIRE(u->B > 0, "advs_unet_forward: no plan (advs_unet_plan first)");
    const size_t img = (size_t)u->S * u->S;
    ADVS_HIP(hipMemcpyAsync(u->x, x_nchw, u->B * u->cfg.in_channels * img * sizeof(float), hipMemcpyDeviceToDevice, u->stream));
    ADVS_HIP(hipMemcpyAsync(u->t, t, u->B * sizeof(long long), hipMemcpyDeviceToDevice, u->stream));
    const int rc = u->fwd.run(u->stream);
    if (rc != ADVS_OK) return rc;
    ADVS_HIP(hipMemcpyAsync(eps_nchw, u->eps, u->B * u->cfg.out_channels * img * sizeof(float), hipMemcpyDeviceToDevice, u->stream));
    return ADVS_OK;
}

// Sampler tables on the host, in double like the reference's torch.float64 tensors (diff_model.py:269-285 schedules, :304 cumprod,
// :428-440 sequences, :450-464 per-step coefficients rounded to f32 where the reference calls .float()).
extern "C" int advs_ddim_tables(int cosine_schedule, int timesteps, int ddim_timesteps, int quad, float eta, float* coef_out,
                                int64_t* tseq_out, int* nsteps_out) {
    ADVS_REQUIRE(timesteps > 0 && ddim_timesteps > 0 && ddim_timesteps <= timesteps && nsteps_out, "advs_ddim_tables: bad step counts");
    std::vector<double> betas(timesteps), ac(timesteps);
    if (cosine_schedule) {
        const double s = 0.008;
        std::vector<double> f(timesteps + 1);
        for (int i = 0; i <= timesteps; ++i) {
            const double x = (double)i;                               // torch.linspace(0, T, T + 1): step 1, exact
            const double cv = cos(((x / timesteps) + s) / (1 + s) * M_PI * 0.5);
            f[i] = cv * cv;
        }
        for (int i = 0; i < timesteps; ++i) {
            double bt = 1 - (f[i + 1] / f[0]) / (f[i] / f[0]);
            betas[i] = bt < 0 ? 0 : (bt > 0.999 ? 0.999 : bt);
        }
    } else {
        // torch.linspace: start + step * i in the first half, end - step * (n - 1 - i) in the second
        const double scale = 1000.0 / timesteps, lo = scale * 0.0001, hi = scale * 0.02;
        const double stp = timesteps > 1 ? (hi - lo) / (timesteps - 1) : 0.0;
        for (int i = 0; i < timesteps; ++i) betas[i] = i < timesteps / 2 ? lo + stp * i : hi - stp * (timesteps - 1 - i);
    }
    double prod = 1.0;
    for (int i = 0; i < timesteps; ++i) { prod *= 1.0 - betas[i]; ac[i] = prod; }
    std::vector<long long> seq;
    if (!quad) {
        const int c = timesteps / ddim_timesteps;
        for (int i = 0; i < timesteps; i += c) seq.push_back(i + 1);
    } else {
        // numpy.linspace: arange(n) * step + start, the last element set to stop
        const double top = sqrt(timesteps * 0.8), stp = ddim_timesteps > 1 ? top / (ddim_timesteps - 1) : 0.0;
        for (int i = 0; i < ddim_timesteps; ++i) {
            const double v = (ddim_timesteps > 1 && i == ddim_timesteps - 1) ? top : i * stp;
            seq.push_back((long long)(v * v) + 1);
        }
    }
    const int n = (int)seq.size();
    *nsteps_out = n;
    if (!coef_out || !tseq_out) return ADVS_OK;                  // size query
    for (int k = 0; k < n; ++k) {
        const int i = n - 1 - k;                                    // loop order: reversed
        const long long ts = seq[i], tp = i ? seq[i - 1] : 0;
        ADVS_REQUIRE(ts >= 0 && ts < timesteps, "advs_ddim_tables: step %lld indexes past the %d-entry schedule (the reference fails the same way)", ts, timesteps);
        const float a_t = (float)ac[ts], a_p = (float)ac[tp];
        const float sigma = eta * sqrtf((1.0f - a_p) / (1.0f - a_t) * (1.0f - a_t / a_p));
        coef_out[3 * k] = a_t; coef_out[3 * k + 1] = a_p; coef_out[3 * k + 2] = sigma;
        tseq_out[k] = ts;
    }
    return ADVS_OK;
}

// The DDIM reverse loop (diff_model.py:442-474, eta = 0 form): x (device, NCHW f32) holds x_T on entry and the sample on return
// (stream order).  coef / tseq: host arrays in LOOP order (advs_ddim_tables).  One captured step = forward + advs_ddim_step.
extern "C" int advs_ddim_run(advs_unet* u, float* x, const float* coef, const int64_t* tseq, int nsteps, int clip_denoised) {
    ADVS_REQUIRE(u && x && coef && tseq && nsteps > 0, "advs_ddim_run: bad arguments");
    ADVS_REQUIRE(u->B > 0 && u->uniform_t, "advs_ddim_run: needs a plan made with uniform_t = 1 (every image of a sampler step sits at one timestep)");
    ADVS_REQUIRE(u->cfg.in_channels == u->cfg.out_channels, "advs_ddim_run: eps and x must have the same shape");
    for (int k = 0; k < nsteps; ++k) ADVS_REQUIRE(coef[3 * k + 2] == 0.f, "advs_ddim_run: sigma != 0 (eta > 0) needs per-step noise, which this entry point does not draw");
    const size_t per = (size_t)u->cfg.in_channels * u->S * u->S;
    if (nsteps > u->nsteps_cap) {
        for (void* p : {(void*)u->coef, (void*)u->tseq}) if (p) (void)hipFree(p);
        ADVS_HIP(hipMalloc((void**)&u->coef, (size_t)nsteps * 3 * sizeof(float)));
        ADVS_HIP(hipMalloc((void**)&u->tseq, (size_t)nsteps * sizeof(long long)));
        if (!u->counter) ADVS_HIP(hipMalloc((void**)&u->counter, sizeof(int)));
        u->nsteps_cap = nsteps;
        u->step.destroy();
    }
    hipStream_t s = u->stream;
    ADVS_HIP(hipMemcpyAsync(u->coef, coef, (size_t)nsteps * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    ADVS_HIP(hipMemcpyAsync(u->tseq, tseq, (size_t)nsteps * sizeof(long long), hipMemcpyHostToDevice, s));
    ADVS_HIP(hipStreamSynchronize(s));                            // the host arrays may go away after the call
    if (u->step.ops.empty() || u->step_clip != (clip_denoised ? 1 : 0) || u->step_nsteps != nsteps) {
        u->step.destroy();
        u->step.ops = u->fwd.ops;
        advs_unet* uu = u;
        const int clip = clip_denoised ? 1 : 0, B = u->B;
        // nsteps is read at replay: the table length is an argument of the launch, so the step is re-captured when it changes
        u->step.ops.push_back([uu, B, per, clip, nsteps](hipStream_t st) {
            return advs_ddim_step(uu->x, uu->eps, nullptr, 0.f, nullptr, uu->coef, (const int64_t*)uu->tseq, nsteps, uu->counter, (int64_t*)uu->t, B, per, clip, st);
        });
        u->step_clip = clip;
        u->step_nsteps = nsteps;
    }
    auto reset = [&]() -> int {
        ADVS_HIP(hipMemsetAsync(u->counter, 0, sizeof(int), s));
        std::vector<long long> t0((size_t)u->B, (long long)tseq[0]);
        ADVS_HIP(hipMemcpyAsync(u->t, t0.data(), u->B * sizeof(long long), hipMemcpyHostToDevice, s));
        ADVS_HIP(hipMemcpyAsync(u->x, x, u->B * per * sizeof(float), hipMemcpyDeviceToDevice, s));
        ADVS_HIP(hipStreamSynchronize(s));
        return ADVS_OK;
    };
    int rc = reset();
    if (rc != ADVS_OK) return rc;
    if (!u->step.graph) {
        if ((rc = u->step.run_eager(s))) return rc;               // validates every launch outside capture
        ADVS_HIP(hipStreamSynchronize(s));
        if ((rc = u->step.capture(s))) return rc;
        if ((rc = reset())) return rc;
    }
    for (int k = 0; k < nsteps; ++k)
        if ((rc = u->step.run(s))) return rc;
    ADVS_HIP(hipMemcpyAsync(x, u->x, u->B * per * sizeof(float), hipMemcpyDeviceToDevice, s));
    return ADVS_OK;
}

extern "C" void advs_unet_destroy(advs_unet* u) {
    if (!u) return;
    drop_plan(u);
    for (void* p : u->wbufs) (void)hipFree(p);
    delete u;
}

// =====================================================================================================================================
// The victim side of the attack loop for the same kind of host: ResNet-50 (timm / torchvision layout: ASR_fast.py:16-20,
// ddim2/diff_model2.py:19-44) with BatchNorm folded into the convs, and the evaluation chain of ASR_fast.py:90-126 in front of it
// (uint8 sampler output -> Pillow-exact Resize((224, 224)) -> ToTensor -> victim -> argmax).  Restates victims.ResNet50 / _ResNetEngine,
// imageops.bilinear_coeffs / resize_u8 / preprocess_batch and asr.evaluate_batch of this package.
// =====================================================================================================================================
struct advs_resnet50 {
    int num_classes = 0, dtype = 0;
    struct Blk { std::string p; int cin, width, cout, stride; bool ds; };
    std::vector<Blk> blocks;
    std::vector<std::pair<std::string, long long>> params;
    std::map<std::string, std::vector<float>> host;
    std::map<std::string, void*> W;
    std::vector<void*> wbufs;
    int stem_kp = 0;
    bool packed = false;
    // plan
    int B = 0, S = 0, src = 0;
    hipStream_t stream = nullptr;
    Arena arena;
    std::list<advs_conv_args> cargs;
    Plan fwd;
    float *x = nullptr, *logits = nullptr;
    // evaluation chain buffers
    unsigned char *hwc = nullptr, *rs1 = nullptr, *rs2 = nullptr;
    int *bounds = nullptr, *coefs = nullptr, ksize = 0;
};

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (libImaging/Resample.c; transforms.Resize on a PIL image,
// ASR_fast.py:94, PSNR_SSIM_fast.py:10-13): bounds [out][2] = (first source pixel, count), coefs [out][ksize] in 22-bit fixed point.
extern "C" int advs_resize_tables(int in_size, int out_size, int* bounds, int* coefs, int* ksize_out) {
    ADVS_REQUIRE(in_size > 0 && out_size > 0 && ksize_out, "advs_resize_tables: bad sizes");
    const double scale = (double)in_size / out_size, filterscale = scale < 1.0 ? 1.0 : scale, support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    *ksize_out = ksize;
    if (!bounds || !coefs) return ADVS_OK;                        // size query
    const double ss = 1.0 / filterscale;
    std::vector<double> w(ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double a = fabs((x + xmin - center + 0.5) * ss);
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < ksize; ++x) {
            double k = x < xmax ? (ww != 0.0 ? w[x] / ww : w[x]) : 0.0;
            coefs[xx * ksize + x] = (int)trunc(k < 0 ? -0.5 + k * (1 << 22) : 0.5 + k * (1 << 22));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ADVS_OK;
}

namespace {

void rn_params(advs_resnet50* r) {
    auto add = [&](const std::string& n, long long k) { r->params.push_back({n, k}); };
    auto bn = [&](const std::string& p, long long c) { add(p + ".weight", c); add(p + ".bias", c); add(p + ".running_mean", c); add(p + ".running_var", c); };
    add("conv1.weight", 64 * 3 * 49);
    bn("bn1", 64);
    const int layers[4][3] = {{64, 3, 1}, {128, 4, 2}, {256, 6, 2}, {512, 3, 2}};
    int cin = 64;
    for (int li = 0; li < 4; ++li)
        for (int bi = 0; bi < layers[li][1]; ++bi) {
            const int width = layers[li][0], s = bi == 0 ? layers[li][2] : 1, cout = width * 4;
            const bool ds = s != 1 || cin != cout;
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
            add(p + ".conv1.weight", (long long)width * cin); bn(p + ".bn1", width);
            add(p + ".conv2.weight", (long long)width * width * 9); bn(p + ".bn2", width);
            add(p + ".conv3.weight", (long long)cout * width); bn(p + ".bn3", cout);
            if (ds) { add(p + ".downsample.0.weight", (long long)cout * cin); bn(p + ".downsample.1", cout); }
            r->blocks.push_back({p, cin, width, cout, s, ds});
            cin = cout;
        }
    add("fc.weight", 2048ll * r->num_classes);
    add("fc.bias", r->num_classes);
}

int rn_dev_copy(advs_resnet50* r, const float* src, size_t n, void** out) {
    void* p = nullptr;
    ADVS_HIP(hipMalloc(&p, n * sizeof(float)));
    r->wbufs.push_back(p);
    ADVS_HIP(hipMemcpy(p, src, n * sizeof(float), hipMemcpyHostToDevice));
    *out = p;
    return ADVS_OK;
}

// victims.ResNet50._fold: w * (gamma / sqrt(var + 1e-5)) per output channel, bias = beta - mean * scale (f32), then pack_conv_weight
// with the input channels padded to whole 128-byte slabs (the 147-column stem)
int rn_fold_pack(advs_resnet50* r, const std::string& conv, const std::string& bnp, int O, int I, int R, const std::string& wkey, const std::string& bkey) {
    const std::vector<float>& w = r->host.at(conv + ".weight");
    const std::vector<float>&g = r->host.at(bnp + ".weight"), &be = r->host.at(bnp + ".bias"), &mu = r->host.at(bnp + ".running_mean"),
                            &var = r->host.at(bnp + ".running_var");
    const int sl = slab(r->dtype), Ip = (I + sl - 1) / sl * sl;
    std::vector<float> wf((size_t)O * Ip * R * R, 0.f), bf(O);
    for (int o = 0; o < O; ++o) {
        const float scale = g[o] / sqrtf(var[o] + 1e-5f);
        bf[o] = be[o] - mu[o] * scale;
        for (int i = 0; i < I; ++i)
            for (int k = 0; k < R * R; ++k) wf[((size_t)o * Ip + i) * R * R + k] = w[((size_t)o * I + i) * R * R + k] * scale;
    }
    void *tmp = nullptr, *dst = nullptr, *bp = nullptr;
    ADVS_HIP(hipMalloc(&tmp, wf.size() * sizeof(float)));
    hipError_t e = hipMemcpy(tmp, wf.data(), wf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&dst, wf.size() * esz(r->dtype));
    int rc = ADVS_OK;
    if (e == hipSuccess) {
        r->wbufs.push_back(dst);
        rc = advs_pack_conv_weight((const float*)tmp, dst, O, Ip, R, R, r->dtype, r->stream);
        if (rc == ADVS_OK) e = hipStreamSynchronize(r->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) ADVS_FAIL(ADVS_ERR_HIP, "resnet50 weights: %s", hipGetErrorString(e));
    if (rc != ADVS_OK) return rc;
    if ((rc = rn_dev_copy(r, bf.data(), bf.size(), &bp))) return rc;
    r->W[wkey] = dst;
    r->W[bkey] = bp;
    return ADVS_OK;
}

int rn_pack(advs_resnet50* r) {
    if (r->packed) return ADVS_OK;
    for (auto& pr : r->params) ADVS_REQUIRE(r->host.count(pr.first), "advs_resnet50_plan: parameter %s was never set", pr.first.c_str());
    int rc;
    // the stem as a GEMM over im2col columns (advs_im2col_nchw): [64][3*7*7 -> whole slabs]
    {
        std::vector<float> w147 = r->host.at("conv1.weight");         // [64][3][7][7] = [64][147] as a 1x1 conv
        r->host["__stem147.weight"] = w147;
        for (const char* s : {".weight", ".bias", ".running_mean", ".running_var"}) r->host[std::string("__stembn") + s] = r->host.at(std::string("bn1") + s);
        if ((rc = rn_fold_pack(r, "__stem147", "__stembn", 64, 147, 1, "stem.wg", "stem.b"))) return rc;
        const int sl = slab(r->dtype);
        r->stem_kp = (147 + sl - 1) / sl * sl;
    }
    for (auto& b : r->blocks) {
        if ((rc = rn_fold_pack(r, b.p + ".conv1", b.p + ".bn1", b.width, b.cin, 1, b.p + ".w1", b.p + ".b1"))) return rc;
        if ((rc = rn_fold_pack(r, b.p + ".conv2", b.p + ".bn2", b.width, b.width, 3, b.p + ".w2", b.p + ".b2"))) return rc;
        if ((rc = rn_fold_pack(r, b.p + ".conv3", b.p + ".bn3", b.cout, b.width, 1, b.p + ".w3", b.p + ".b3"))) return rc;
        if (b.ds && (rc = rn_fold_pack(r, b.p + ".downsample.0", b.p + ".downsample.1", b.cout, b.cin, 1, b.p + ".wd", b.p + ".bd"))) return rc;
    }
    void* q = nullptr;
    if ((rc = rn_dev_copy(r, r->host.at("fc.weight").data(), r->host.at("fc.weight").size(), &q))) return rc;
    r->W["fc.w"] = q;
    if ((rc = rn_dev_copy(r, r->host.at("fc.bias").data(), r->host.at("fc.bias").size(), &q))) return rc;
    r->W["fc.b"] = q;
    r->packed = true;
    return ADVS_OK;
}

void rn_drop_plan(advs_resnet50* r) {
    r->fwd.destroy();
    r->arena.destroy();
    r->cargs.clear();
    for (void* p : {(void*)r->x, (void*)r->logits, (void*)r->hwc, (void*)r->rs1, (void*)r->rs2, (void*)r->bounds, (void*)r->coefs})
        if (p) (void)hipFree(p);
    r->x = r->logits = nullptr; r->hwc = r->rs1 = r->rs2 = nullptr; r->bounds = r->coefs = nullptr;
    r->B = r->S = r->src = 0;
}

// _ResNetEngine.__init__ (this package's victims.py)
int rn_emit(advs_resnet50* r) {
    int rc = ADVS_OK;
    const int B = r->B, S = r->S, dt = r->dtype;
    auto alloc = [&](size_t bytes) { void* p = nullptr; if (rc == ADVS_OK) rc = r->arena.alloc(bytes, &p); return p; };
    auto buf = [&](int H, int W, int C) { Act a; a.p = alloc((size_t)B * H * W * C * esz(dt)); a.B = B; a.H = H; a.W = W; a.C = C; return a; };
    auto conv = [&](const Act& x, const std::string& wk, const std::string& bk, int cout, int ksize, int stride, int pad, int act, const Act* res) {
        const int Ho = (x.H + 2 * pad - ksize) / stride + 1, Wo = (x.W + 2 * pad - ksize) / stride + 1;
        Act y = buf(Ho, Wo, cout);
        r->cargs.emplace_back();
        advs_conv_args& a = r->cargs.back();
        memset(&a, 0, sizeof(a));
        a.x1 = x.p; a.w = r->W.at(wk); a.bias = (const float*)r->W.at(bk); a.residual = res ? res->p : nullptr; a.y = y.p;
        a.b = B; a.h = x.H; a.w_ = x.W; a.c1 = x.C; a.cout = cout; a.ksize = ksize; a.stride = stride; a.pad = pad; a.act = act; a.dtype = dt;
        const advs_conv_args* ap = &a;
        r->fwd.ops.push_back([ap](hipStream_t s) { return advs_conv2d(ap, s); });
        return y;
    };
    const int ho = (S + 6 - 7) / 2 + 1, kp = r->stem_kp;
    Act cols = buf(ho, ho, kp);
    {
        const float* xp = r->x;
        void* cp = cols.p;
        r->fwd.ops.push_back([=](hipStream_t s) { return advs_im2col_nchw(xp, cp, B, 3, S, S, 7, 2, 3, kp, dt, s); });
    }
    Act h = conv(cols, "stem.wg", "stem.b", 64, 1, 1, 0, ADVS_ACT_RELU, nullptr);
    r->arena.release(cols.p);
    const int hp = (ho + 2 - 3) / 2 + 1;
    Act pooled = buf(hp, hp, 64);
    {
        const void* hpn = h.p;
        void* pp = pooled.p;
        r->fwd.ops.push_back([=](hipStream_t s) { return advs_maxpool3x3s2(hpn, pp, B, ho, ho, 64, dt, s); });
    }
    r->arena.release(h.p);
    h = pooled;
    for (auto& b : r->blocks) {
        Act y1 = conv(h, b.p + ".w1", b.p + ".b1", b.width, 1, 1, 0, ADVS_ACT_RELU, nullptr);
        Act y2 = conv(y1, b.p + ".w2", b.p + ".b2", b.width, 3, b.stride, 1, ADVS_ACT_RELU, nullptr);
        r->arena.release(y1.p);
        Act idn = b.ds ? conv(h, b.p + ".wd", b.p + ".bd", b.cout, 1, b.stride, 0, ADVS_ACT_NONE, nullptr) : h;
        Act y3 = conv(y2, b.p + ".w3", b.p + ".b3", b.cout, 1, 1, 0, ADVS_ACT_RELU, &idn);
        r->arena.release(y2.p);
        if (b.ds) r->arena.release(idn.p);
        r->arena.release(h.p);
        h = y3;
    }
    float* avg = (float*)alloc((size_t)B * h.C * sizeof(float));
    {
        const void* hpn = h.p;
        const int hw = h.H * h.W, C = h.C, nc = r->num_classes;
        const float *fw = (const float*)r->W.at("fc.w"), *fb = (const float*)r->W.at("fc.b");
        float* lg = r->logits;
        r->fwd.ops.push_back([=](hipStream_t s) { return advs_global_avgpool(hpn, avg, B, hw, C, dt, s); });
        r->fwd.ops.push_back([=](hipStream_t s) { return advs_linear_f32(avg, fw, fb, lg, B, C, nc, ADVS_ACT_NONE, ADVS_ACT_NONE, s); });
    }
    return rc;
}

}  // namespace

extern "C" int advs_resnet50_create(int num_classes, int dtype, advs_resnet50** out) {
    ADVS_REQUIRE(out && num_classes > 0, "advs_resnet50_create: bad arguments");
    ADVS_REQUIRE(dtype == ADVS_F32 || dtype == ADVS_BF16 || dtype == ADVS_F16, "advs_resnet50_create: unknown dtype %d", dtype);
    advs_resnet50* r = new advs_resnet50();
    r->num_classes = num_classes;
    r->dtype = dtype;
    rn_params(r);
    *out = r;
    return ADVS_OK;
}
extern "C" int advs_resnet50_param_count(const advs_resnet50* r) { return r ? (int)r->params.size() : 0; }
extern "C" int advs_resnet50_param_name(const advs_resnet50* r, int i, char* name, int name_len, long long* numel) {
    ADVS_REQUIRE(r && i >= 0 && i < (int)r->params.size() && name && name_len > 0, "advs_resnet50_param_name: bad index");
    ADVS_REQUIRE((int)r->params[i].first.size() < name_len, "advs_resnet50_param_name: %d bytes are too few for %s", name_len, r->params[i].first.c_str());
    snprintf(name, (size_t)name_len, "%s", r->params[i].first.c_str());
    if (numel) *numel = r->params[i].second;
    return ADVS_OK;
}
extern "C" int advs_resnet50_set_param(advs_resnet50* r, const char* name, const float* host_data, long long numel) {
    ADVS_REQUIRE(r && name && host_data, "advs_resnet50_set_param: null pointer");
    long long want = -1;
    for (auto& pr : r->params)
        if (pr.first == name) want = pr.second;
    ADVS_REQUIRE(want >= 0, "advs_resnet50_set_param: %s is not a parameter of this network (BatchNorm's num_batches_tracked is not one)", name);
    ADVS_REQUIRE(numel == want, "advs_resnet50_set_param: %s has %lld elements, got %lld", name, want, numel);
    r->host[name].assign(host_data, host_data + numel);
    if (r->packed) {
        rn_drop_plan(r);
        for (void* p : r->wbufs) (void)hipFree(p);
        r->wbufs.clear(); r->W.clear(); r->packed = false;
    }
    return ADVS_OK;
}

// batch images of size x size into the victim; src_size > 0 also prepares advs_resnet50_eval_u8's chain for src_size x src_size uint8 images
extern "C" int advs_resnet50_plan(advs_resnet50* r, int batch, int size, int src_size, void* stream) {
    ADVS_REQUIRE(r && batch > 0 && size >= 32 && src_size >= 0 && stream, "advs_resnet50_plan: bad arguments (an explicit stream is needed)");
    int rc = advs_init();
    if (rc != ADVS_OK) return rc;
    if (r->B) rn_drop_plan(r);
    r->stream = (hipStream_t)stream;
    if ((rc = rn_pack(r))) return rc;
    r->B = batch; r->S = size; r->src = src_size;
    ADVS_HIP(hipMalloc((void**)&r->x, (size_t)batch * 3 * size * size * sizeof(float)));
    ADVS_HIP(hipMalloc((void**)&r->logits, (size_t)batch * r->num_classes * sizeof(float)));
    ADVS_HIP(hipMemset(r->x, 0, (size_t)batch * 3 * size * size * sizeof(float)));
    if (src_size > 0) {
        ADVS_HIP(hipMalloc((void**)&r->hwc, (size_t)batch * src_size * src_size * 3));
        if (src_size != size) {
            ADVS_HIP(hipMalloc((void**)&r->rs1, (size_t)batch * src_size * size * 3));
            ADVS_HIP(hipMalloc((void**)&r->rs2, (size_t)batch * size * size * 3));
            int ks = 0;
            if ((rc = advs_resize_tables(src_size, size, nullptr, nullptr, &ks))) { rn_drop_plan(r); return rc; }
            std::vector<int> bd(2 * size), cf((size_t)size * ks);
            if ((rc = advs_resize_tables(src_size, size, bd.data(), cf.data(), &ks))) { rn_drop_plan(r); return rc; }
            r->ksize = ks;
            ADVS_HIP(hipMalloc((void**)&r->bounds, bd.size() * sizeof(int)));
            ADVS_HIP(hipMalloc((void**)&r->coefs, cf.size() * sizeof(int)));
            ADVS_HIP(hipMemcpy(r->bounds, bd.data(), bd.size() * sizeof(int), hipMemcpyHostToDevice));
            ADVS_HIP(hipMemcpy(r->coefs, cf.data(), cf.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    if ((rc = rn_emit(r))) { rn_drop_plan(r); return rc; }
    if ((rc = r->fwd.run_eager(r->stream))) { rn_drop_plan(r); return rc; }
    ADVS_HIP(hipStreamSynchronize(r->stream));
    if ((rc = r->fwd.capture(r->stream))) { rn_drop_plan(r); return rc; }
    return ADVS_OK;
}

// logits = model(x): x [B][3][S][S] f32, logits [B][num_classes] f32, device pointers, stream-ordered (ASR_fast.py:113-115)
extern "C" int advs_resnet50_forward(advs_resnet50* r, const float* x_nchw, float* logits) {
    ADVS_REQUIRE(r && x_nchw && logits, "advs_resnet50_forward: null pointer");
    ADVS_REQUIRE(r->B > 0, "advs_resnet50_forward: no plan (advs_resnet50_plan first)");
    ADVS_HIP(hipMemcpyAsync(r->x, x_nchw, (size_t)r->B * 3 * r->S * r->S * sizeof(float), hipMemcpyDeviceToDevice, r->stream));
    const int rc = r->fwd.run(r->stream);
    if (rc != ADVS_OK) return rc;
    ADVS_HIP(hipMemcpyAsync(logits, r->logits, (size_t)r->B * r->num_classes * sizeof(float), hipMemcpyDeviceToDevice, r->stream));
    return ADVS_OK;
}

// asr.evaluate_batch (ASR_fast.py:90-97, 113-117): uint8 [B][3][src][src] sampler output -> HWC -> Resize((S, S)) (Pillow BILINEAR, horizontal
// pass first) -> ToTensor -> victim -> argmax.  pred: int32 [B] on the device.  Stream-ordered.
extern "C" int advs_resnet50_eval_u8(advs_resnet50* r, const uint8_t* images_nchw, int* pred) {
    ADVS_REQUIRE(r && images_nchw && pred, "advs_resnet50_eval_u8: null pointer");
    ADVS_REQUIRE(r->B > 0 && r->src > 0, "advs_resnet50_eval_u8: needs a plan made with src_size > 0");
    void* s = r->stream;
    const int B = r->B, S = r->S, src = r->src;
    int rc;
    if ((rc = advs_u8_nchw_to_hwc(images_nchw, r->hwc, B, 3, src, src, s))) return rc;
    const unsigned char* img = r->hwc;
    if (src != S) {
        if ((rc = advs_resample_u8(r->hwc, r->rs1, r->bounds, r->coefs, r->ksize, B, src, src, src, S, 3, 1, s))) return rc;
        if ((rc = advs_resample_u8(r->rs1, r->rs2, r->bounds, r->coefs, r->ksize, B, src, S, S, S, 3, 0, s))) return rc;
        img = r->rs2;
    }
    if ((rc = advs_u8hwc_to_f32nchw(img, r->x, B, S, S, 3, nullptr, nullptr, s))) return rc;
    if ((rc = r->fwd.run(r->stream))) return rc;
    return advs_argmax_rows(r->logits, pred, B, r->num_classes, s);
}

extern "C" void advs_resnet50_destroy(advs_resnet50* r) {
    if (!r) return;
    rn_drop_plan(r);
    for (void* p : r->wbufs) (void)hipFree(p);
    delete r;
}
