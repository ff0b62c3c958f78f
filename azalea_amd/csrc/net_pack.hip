// net_pack.hip -- HexNetwork weights -> the packed operands net_kernels.hip's towers and heads read.
//
// What is packed (Network / HexNetwork parameters, azalea/network.py:42-61, :120-132): eval-mode BatchNorm folded
// into the convolution before it (scale = w / sqrt(var + 1e-5), shift = b - mean * scale, in f64), the 3 -> 4
// embedding folded through the stem convolution into a [tap][cell value][cout] table, the 3x3 filters re-ordered
// into the MFMA fragment order of the tower the engine launches (fp32 B fragments, or hi+lo f16 halves for the
// split-f16 towers), the FC layers transposed / tiled for the heads kernels.
//
// Two implementations of the same arithmetic:
//   * device (default): kernels on the engine stream that read the LIVE state_dict tensors in place (torch
//     `data_ptr()`s; host arrays are staged through one arena first) and write the persistent packed buffers --
//     the trainer changes its weights every step (policy_trainer.py:85-90) and refreshes the engine on every
//     Player.read, so no tensor makes a host round trip;
//   * host (AZX_PACK=host, read once per network): the scalar reference loops, kept as the checker --
//     tests/test_gpu_weights.py holds the two bit-identical on every packed buffer.
// Range guard: a folded weight beyond the f16 range cannot be split into hi+lo halves; both paths report the
// largest |folded weight| per tensor and azx_net_set_weights rejects such a network (AZX_ERANGE) by name.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "net_priv.h"

#define F16_MAX 65504.0f

// ---- raw tensor table: fixed slots, the same on the host and on the device -------------------------------------
enum {
    RT_EMB = 0, RT_CONV1 = 1, RT_BN1 = 2 /* weight, bias, running_mean, running_var */,
    RT_VCONV = 6, RT_VBN = 7, RT_PCONV = 11, RT_PBN = 12,
    RT_FC2W = 16, RT_FC2B = 17, RT_FC3W = 18, RT_FC3B = 19, RT_MFW = 20, RT_MFB = 21,
    RT_LAYER0 = 22 /* per tower layer: conv weight, bn weight, bias, running_mean, running_var */
};

struct RawSlot { std::string name; size_t count; };

static std::vector<RawSlot> raw_slots(int C, int n2, int L) {
    std::vector<RawSlot> s(RT_LAYER0 + 5 * (size_t)L);
    static const char *bn[4] = {".weight", ".bias", ".running_mean", ".running_var"};
    s[RT_EMB] = {"encoder.weight", 12};
    s[RT_CONV1] = {"conv1.weight", (size_t)C * 36};
    for (int i = 0; i < 4; ++i) {
        s[RT_BN1 + i] = {std::string("bn1") + bn[i], (size_t)C};
        s[RT_VBN + i] = {std::string("value_bn1") + bn[i], 2};
        s[RT_PBN + i] = {std::string("move_bn1") + bn[i], 4};
    }
    s[RT_VCONV] = {"value_conv1.weight", (size_t)2 * C};
    s[RT_PCONV] = {"move_conv1.weight", (size_t)4 * C};
    s[RT_FC2W] = {"value_fc2.weight", (size_t)64 * 2 * n2};
    s[RT_FC2B] = {"value_fc2.bias", 64};
    s[RT_FC3W] = {"value_fc3.weight", 64};
    s[RT_FC3B] = {"value_fc3.bias", 1};
    s[RT_MFW] = {"move_fc.weight", (size_t)n2 * 4 * n2};
    s[RT_MFB] = {"move_fc.bias", (size_t)n2};
    for (int l = 0; l < L; ++l) {
        char nm[96];
        snprintf(nm, sizeof nm, "resblocks.%d.conv%d.weight", l / 2, l % 2 + 1);
        s[RT_LAYER0 + 5 * l] = {nm, (size_t)C * C * 9};
        for (int i = 0; i < 4; ++i) {
            snprintf(nm, sizeof nm, "resblocks.%d.bn%d%s", l / 2, l % 2 + 1, bn[i]);
            s[RT_LAYER0 + 5 * l + 1 + i] = {nm, (size_t)C};
        }
    }
    return s;
}

// ---- the persistent packed buffers -----------------------------------------------------------------------------
struct Plan {
    int C, N, n2, L, NT, NT16, NCH, KVp, KPp, NTP;
    bool wg, wp, s32, s16, hd16;          // which tower operands this network's kernels read
    size_t n_stemT, n_Wg, n_Wp, n_Ws, n_Ws16, n_Wh16, n_Whd16, n_hmP, n_hmV, n_fc2T, n_mfcT;
};

static Plan make_plan(const AzxNet *net) {
    Plan p;
    p.C = net->d.C; p.N = net->d.N; p.n2 = p.N * p.N; p.L = net->d.layers;
    p.NT = p.C / 32; p.NT16 = p.C / 16; p.NCH = p.C / 64;
    p.KVp = (2 * p.n2 + 15) & ~15; p.KPp = (4 * p.n2 + 15) & ~15; p.NTP = (p.n2 + 15) / 16;
    const int v = net->tower_variant;
    const bool f16 = v == 4 || v == 5;
    p.wg = v == 0;                                     // k_conv_generic
    p.wp = v >= 1 && v <= 3;                           // k_tower_mfma (fp32)
    p.s32 = v == 5;                                    // the wide stem (k_stem_wide_f16x3) reads its table in 32x32x16 fragment order
    p.s16 = f16;                                       // 16x16x32 fragment order: both split-f16 towers
    p.hd16 = v == 4 && p.C == 64;
    p.n_stemT = (size_t)28 * p.C;
    p.n_Wg = (size_t)p.L * 9 * p.C * p.C;
    p.n_Wp = p.n_Wg;
    p.n_Ws = (size_t)2 * p.NT * 2 * 64 * 8;
    p.n_Ws16 = (size_t)p.NT16 * 2 * 64 * 8;
    p.n_Wh16 = (size_t)p.L * 9 * p.NCH * 2 * p.NT16 * 2 * 64 * 8;
    p.n_Whd16 = (size_t)2 * 2 * 64 * 8;
    p.n_hmP = (size_t)p.NTP * (p.KPp / 16) * 64 * 4;
    p.n_hmV = (size_t)4 * (p.KVp / 16) * 64 * 4;
    p.n_fc2T = (size_t)2 * p.n2 * 64;
    p.n_mfcT = (size_t)4 * p.n2 * AZX_CELL_STRIDE;
    return p;
}

template <typename T>
static bool pbuf(AzxNet *net, const char *name, const T *&slot, size_t count) {
    T *p = nalloc<T>(net, count);          // zero-filled
    if (!p) return false;
    slot = p;
    net->packs.push_back({name, (void *)p, count * sizeof(T)});
    return true;
}

static int ensure_buffers(AzxNet *net, const Plan &p) {
    if (net->packed_once) return AZX_OK;
    NetDev &d = net->d;
    bool ok = pbuf(net, "stemT", d.stemT, p.n_stemT) && pbuf(net, "stem_b", d.stem_b, p.C) &&
              pbuf(net, "bias", d.bias, (size_t)std::max(1, p.L) * p.C) &&
              pbuf(net, "wv", d.wv, (size_t)2 * p.C) && pbuf(net, "bv", d.bv, 2) &&
              pbuf(net, "wp", d.wp, (size_t)4 * p.C) && pbuf(net, "bp", d.bp, 4) &&
              pbuf(net, "fc2T", d.fc2T, p.n_fc2T) && pbuf(net, "fc2b", d.fc2b, 64) &&
              pbuf(net, "fc3w", d.fc3w, 64) && pbuf(net, "fc3b", d.fc3b, 1) &&
              pbuf(net, "mfcT", d.mfcT, p.n_mfcT) && pbuf(net, "mfcb", d.mfcb, AZX_CELL_STRIDE) &&
              pbuf(net, "hmP", d.hmP, p.n_hmP) && pbuf(net, "hmV", d.hmV, p.n_hmV);
    if (ok && p.wg) ok = pbuf(net, "Wg", d.Wg, p.n_Wg);
    if (ok && p.wp) ok = pbuf(net, "Wp", d.Wp, p.n_Wp);
    if (ok && p.s32) ok = pbuf(net, "Ws", d.Ws, p.n_Ws);
    if (ok && p.s16) ok = pbuf(net, "Ws16", d.Ws16, p.n_Ws16) && pbuf(net, "Wh16", d.Wh16, p.n_Wh16);
    if (ok && p.hd16) ok = pbuf(net, "Whd16", d.Whd16, p.n_Whd16) && pbuf(net, "hbias16", d.hbias16, 16);
    if (!ok) return azx_net_fail(AZX_ENOMEM, "net: hipMalloc of the packed weight buffers failed");
    d.hm_lda = ((p.KVp + p.KPp - 4 + 63) / 64) * 64 + 4;        // smallest stride = 4 (mod 64) that holds a row
    const int nf = p.C * (p.L + 1) + 6;
    net->fold = nalloc<double>(net, (size_t)2 * nf);
    net->wmax = nalloc<uint32_t>(net, (size_t)p.L + 2);
    net->raw_slots = RT_LAYER0 + 5 * p.L;
    net->raw_tab = reinterpret_cast<const float **>(nalloc<const float *>(net, (size_t)net->raw_slots));
    if (!net->fold || !net->wmax || !net->raw_tab ||
        hipHostMalloc((void **)&net->raw_tab_host, sizeof(const float *) * net->raw_slots) != hipSuccess ||
        hipHostMalloc((void **)&net->wmax_host, sizeof(uint32_t) * (p.L + 2)) != hipSuccess)
        return azx_net_fail(AZX_ENOMEM, "net: allocating the pack scratch failed");
    // nalloc zero-fills with hipMemset on the null stream, which the engine's non-blocking stream does not wait for:
    // let the fills land before anything is packed into the buffers
    if (hipDeviceSynchronize() != hipSuccess) return azx_net_fail(AZX_EHIP, "net: device sync after allocating the packed buffers failed");
    net->packed_once = true;
    return AZX_OK;
}

static const void *pack_ptr(const AzxNet *net, const char *name) {
    for (const auto &b : net->packs)
        if (b.name == name) return b.ptr;
    return nullptr;
}

// =================================================================================================================
// host reference pack (AZX_PACK=host)
// =================================================================================================================
static inline unsigned short f16bits_host(float w, int part) {
    const _Float16 hi = (_Float16)w;
    const _Float16 lo = (_Float16)(w - (float)hi);
    const _Float16 v = part ? lo : hi;
    unsigned short bits;
    memcpy(&bits, &v, 2);
    return part ? lo_round_bits(bits) : bits;
}

static int pack_host(AzxNet *net, const Plan &plan, const std::map<std::string, std::vector<float>> &T, float *wmax_out) {
    const int C = net->d.C, N = net->d.N, n2 = N * N, L = net->d.layers;
    std::string missing;
    auto get = [&](const std::string &name, size_t want) -> const std::vector<float> * {
        auto it = T.find(name);
        if (it == T.end() || it->second.size() != want) {
            missing = "net: tensor '" + name + "' missing or has the wrong size";
            return nullptr;
        }
        return &it->second;
    };
    // eval-mode BatchNorm2d folded to scale/shift, eps 1e-5 (network.py:21,:48)
    auto fold = [&](const std::string &pre, int c, std::vector<double> &scale, std::vector<double> &shift) -> bool {
        auto w = get(pre + ".weight", c), b = get(pre + ".bias", c), m = get(pre + ".running_mean", c),
             v = get(pre + ".running_var", c);
        if (!w || !b || !m || !v) return false;
        scale.resize(c);
        shift.resize(c);
        for (int i = 0; i < c; ++i) {
            scale[i] = (double)(*w)[i] / std::sqrt((double)(*v)[i] + 1e-5);
            shift[i] = (double)(*b)[i] - (double)(*m)[i] * scale[i];
        }
        return true;
    };
#define NEED(x) if (!(x)) return azx_net_fail(AZX_EINVAL, missing.c_str())
    std::vector<double> sc, sh;
    // stem table: T[tap][v][co] = scale[co] * sum_i emb[v][i] * w[co][i][tap]
    auto emb = get("encoder.weight", 12);
    auto w1 = get("conv1.weight", (size_t)C * 4 * 9);
    NEED(emb && w1 && fold("bn1", C, sc, sh));
    std::vector<float> stemT((size_t)(9 * 3 + 1) * C, 0.0f), stem_b(C);   // row 27: zeros (off-board taps)
    for (int tap = 0; tap < 9; ++tap)
        for (int v = 0; v < 3; ++v)
            for (int co = 0; co < C; ++co) {
                double s = 0;
                for (int i = 0; i < 4; ++i) s += (double)(*emb)[v * 4 + i] * (double)(*w1)[(co * 4 + i) * 9 + tap];
                stemT[(tap * 3 + v) * C + co] = (float)(s * sc[co]);
            }
    for (int co = 0; co < C; ++co) stem_b[co] = (float)sh[co];
    // tower
    std::vector<float> Wg((size_t)L * 9 * C * C), bias((size_t)L * C);
    for (int l = 0; l < L; ++l) {
        char nm[128];
        snprintf(nm, sizeof nm, "resblocks.%d.conv%d.weight", l / 2, l % 2 + 1);
        auto w = get(nm, (size_t)C * C * 9);
        snprintf(nm, sizeof nm, "resblocks.%d.bn%d", l / 2, l % 2 + 1);
        NEED(w && fold(nm, C, sc, sh));
        for (int tap = 0; tap < 9; ++tap)
            for (int ci = 0; ci < C; ++ci)
                for (int co = 0; co < C; ++co)
                    Wg[(((size_t)l * 9 + tap) * C + ci) * C + co] =
                        (float)((double)(*w)[((size_t)co * C + ci) * 9 + tap] * sc[co]);
        for (int co = 0; co < C; ++co) bias[(size_t)l * C + co] = (float)sh[co];
    }
    std::vector<float> Wp;
    if (net->use_mfma && net->tower_variant < 4) {
        // B-fragment order: [layer][tap][q][ntile][lane(j + 32 h)][t] = W[tap][cin 8q+4h+t][cout 32 ntile + j]
        const int NT = C / 32, Q = C / 8;
        Wp.resize((size_t)L * 9 * Q * NT * 64 * 4);
        for (int l = 0; l < L; ++l)
            for (int tap = 0; tap < 9; ++tap)
                for (int q = 0; q < Q; ++q)
                    for (int nt = 0; nt < NT; ++nt)
                        for (int ln = 0; ln < 64; ++ln)
                            for (int t = 0; t < 4; ++t) {
                                const int j = ln & 31, h = ln >> 5;
                                const int ci = 8 * q + 4 * h + t, co = 32 * nt + j;
                                Wp[(((((size_t)l * 9 + tap) * Q + q) * NT + nt) * 64 + ln) * 4 + t] =
                                    Wg[(((size_t)l * 9 + tap) * C + ci) * C + co];
                            }
    }
    std::vector<unsigned short> Ws;
    if (net->tower_variant == 5) {
        const int NT = C / 32;
        auto f16bits = [](float w, int part) -> unsigned short {
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)(w - (float)hi);
            const _Float16 v = part ? lo : hi;
            unsigned short bits;
            memcpy(&bits, &v, 2);
            return part ? lo_round_bits(bits) : bits;
        };
        // stem table as K = 27 (tap*3 + colour, padded to 32) x C weights:
        // [kk][ntile][part hi/lo][lane j + 32 h][t] = split(stemT[k = 16 kk + 8 h + t][cout 32 ntile + j])
        Ws.resize((size_t)2 * NT * 2 * 64 * 8);
        size_t os = 0;
        for (int kk = 0; kk < 2; ++kk)
            for (int nt = 0; nt < NT; ++nt)
                for (int part = 0; part < 2; ++part)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int t = 0; t < 8; ++t) {
                            const int j = ln & 31, h = ln >> 5;
                            const int k = 16 * kk + 8 * h + t, co = 32 * nt + j;
                            Ws[os++] = f16bits(k < 27 ? stemT[(size_t)k * C + co] : 0.0f, part);
                        }
    }
    std::vector<unsigned short> Wh16, Ws16;
    if (net->tower_variant == 4 || net->tower_variant == 5) {
        const int NT16 = C / 16, NCH = C / 64;
        auto f16bits = [](float w, int part) -> unsigned short {
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)(w - (float)hi);
            const _Float16 v = part ? lo : hi;
            unsigned short bits;
            memcpy(&bits, &v, 2);
            return part ? lo_round_bits(bits) : bits;
        };
        // 16x16x32 A-operand order: lane (j = lane & 15: output channel in the tile, h = lane >> 4: k-group)
        // holds 8 consecutive k.  Stem: [ntile][hi,lo][lane][t] = split(stemT[k = 8 h + t][cout 16 ntile + j])
        Ws16.resize((size_t)NT16 * 2 * 64 * 8);
        size_t os = 0;
        for (int nt = 0; nt < NT16; ++nt)
            for (int part = 0; part < 2; ++part)
                for (int ln = 0; ln < 64; ++ln)
                    for (int t = 0; t < 8; ++t) {
                        const int j = ln & 15, h = ln >> 4, k = 8 * h + t, co = 16 * nt + j;
                        Ws16[os++] = f16bits(k < 27 ? stemT[(size_t)k * C + co] : 0.0f, part);
                    }
        // convs, one 32-channel k-step after the other:
        // [layer][tap][64-channel chunk][half][ntile][hi,lo][lane][t]
        //   = split(W[tap][cin 64 chunk + 32 half + 8 h + t][cout 16 ntile + j])
        Wh16.resize((size_t)L * 9 * NCH * 2 * NT16 * 2 * 64 * 8);
        size_t o = 0;
        for (int l = 0; l < L; ++l)
            for (int tap = 0; tap < 9; ++tap)
                for (int ch = 0; ch < NCH; ++ch)
                    for (int half = 0; half < 2; ++half)
                        for (int nt = 0; nt < NT16; ++nt)
                            for (int part = 0; part < 2; ++part)
                                for (int ln = 0; ln < 64; ++ln)
                                    for (int t = 0; t < 8; ++t) {
                                        const int j = ln & 15, h = ln >> 4;
                                        const int ci = 64 * ch + 32 * half + 8 * h + t;
                                        int co = 16 * nt + j;
                                        // wide tower: row j = 4 lh + r of tile n = nt % 4 of a wave's 64-channel
                                        // group is channel 32 (n >> 1) + 8 lh + 4 (n & 1) + r of the group, so a
                                        // lane's accumulators are 8 consecutive channels per tile pair
                                        // (k_conv_wide_f16x3_s16's 16-byte epilogue pieces)
                                        if (net->tower_variant == 5)
                                            co = 64 * (nt / 4) + 32 * ((nt % 4) >> 1) + 8 * (j >> 2) + 4 * (nt & 1) + (j & 3);
                                        Wh16[o++] = f16bits(Wg[(((size_t)l * 9 + tap) * C + ci) * C + co], part);
                                    }
    }
    // heads
    auto wvc = get("value_conv1.weight", (size_t)2 * C), wpc = get("move_conv1.weight", (size_t)4 * C);
    std::vector<double> scv, shv, scp, shp;
    NEED(wvc && wpc && fold("value_bn1", 2, scv, shv) && fold("move_bn1", 4, scp, shp));
    std::vector<float> wv((size_t)2 * C), bv(2), wp((size_t)4 * C), bp(4);
    for (int o = 0; o < 2; ++o) {
        for (int c = 0; c < C; ++c) wv[(size_t)o * C + c] = (float)((double)(*wvc)[(size_t)o * C + c] * scv[o]);
        bv[o] = (float)shv[o];
    }
    for (int o = 0; o < 4; ++o) {
        for (int c = 0; c < C; ++c) wp[(size_t)o * C + c] = (float)((double)(*wpc)[(size_t)o * C + c] * scp[o]);
        bp[o] = (float)shp[o];
    }
    std::vector<unsigned short> Whd16;
    std::vector<float> hbias16(16, 0.f);
    if (net->tower_variant == 4 && C == 64) {
        auto f16bits = [](float w, int part) -> unsigned short {
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)(w - (float)hi);
            const _Float16 v = part ? lo : hi;
            unsigned short bits;
            memcpy(&bits, &v, 2);
            return part ? lo_round_bits(bits) : bits;
        };
        Whd16.resize((size_t)2 * 2 * 64 * 8);
        size_t oh = 0;
        for (int ks = 0; ks < 2; ++ks)
            for (int part = 0; part < 2; ++part)
                for (int ln = 0; ln < 64; ++ln)
                    for (int t = 0; t < 8; ++t) {
                        const int o = ln & 15, ci = 32 * ks + 8 * (ln >> 4) + t;
                        const float w = o < 2 ? wv[(size_t)o * C + ci] : o < 6 ? wp[(size_t)(o - 2) * C + ci] : 0.0f;
                        Whd16[oh++] = f16bits(w, part);
                    }
        for (int o = 0; o < 6; ++o) hbias16[o] = o < 2 ? bv[o] : bp[o - 2];
    }
    auto fc2w = get("value_fc2.weight", (size_t)64 * 2 * n2), fc2b = get("value_fc2.bias", 64);
    auto fc3w = get("value_fc3.weight", 64), fc3b = get("value_fc3.bias", 1);
    auto mfw = get("move_fc.weight", (size_t)n2 * 4 * n2), mfb = get("move_fc.bias", n2);
    NEED(fc2w && fc2b && fc3w && fc3b && mfw && mfb);
    std::vector<float> fc2T((size_t)2 * n2 * 64), mfcT((size_t)4 * n2 * AZX_CELL_STRIDE, 0.f), mfcb(AZX_CELL_STRIDE, 0.f);
    for (int o = 0; o < 64; ++o)
        for (int i = 0; i < 2 * n2; ++i) fc2T[(size_t)i * 64 + o] = (*fc2w)[(size_t)o * 2 * n2 + i];
    for (int t = 0; t < n2; ++t) {
        for (int i = 0; i < 4 * n2; ++i) mfcT[(size_t)i * AZX_CELL_STRIDE + t] = (*mfw)[(size_t)t * 4 * n2 + i];
        mfcb[t] = (*mfb)[t];
    }
    // k_heads_mfma's B operands: [n tile of 16][k group t of 16][lane][s] = W[k = 16 t + 4 (lane >> 4) + s][unit 16 tile + (lane & 15)]
    const int KVp = (2 * n2 + 15) & ~15, KPp = (4 * n2 + 15) & ~15, NTP = (n2 + 15) / 16;
    std::vector<float> hmP((size_t)NTP * (KPp / 16) * 64 * 4, 0.f), hmV((size_t)4 * (KVp / 16) * 64 * 4, 0.f);
    for (int t = 0; t < NTP; ++t)
        for (int q = 0; q < KPp / 16; ++q)
            for (int l = 0; l < 64; ++l)
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int k = 16 * q + 4 * (l >> 4) + sidx, unit = 16 * t + (l & 15);
                    if (k < 4 * n2 && unit < n2)
                        hmP[(((size_t)t * (KPp / 16) + q) * 64 + l) * 4 + sidx] = (*mfw)[(size_t)unit * 4 * n2 + k];
                }
    for (int t = 0; t < 4; ++t)
        for (int q = 0; q < KVp / 16; ++q)
            for (int l = 0; l < 64; ++l)
                for (int sidx = 0; sidx < 4; ++sidx) {
                    const int k = 16 * q + 4 * (l >> 4) + sidx, unit = 16 * t + (l & 15);
                    if (k < 2 * n2) hmV[(((size_t)t * (KVp / 16) + q) * 64 + l) * 4 + sidx] = (*fc2w)[(size_t)unit * 2 * n2 + k];
                }
#undef NEED
    // largest |folded weight| per group: stem table, tower layers, head convs
    auto amax = [](const float *p, size_t n) { float m = 0.f; for (size_t i = 0; i < n; ++i) { const float a = std::fabs(p[i]); if (!(a <= m)) m = a; } return m; };
    wmax_out[0] = amax(stemT.data(), stemT.size());
    for (int l = 0; l < L; ++l) wmax_out[1 + l] = amax(Wg.data() + (size_t)l * 9 * C * C, (size_t)9 * C * C);
    wmax_out[1 + L] = std::max(amax(wv.data(), wv.size()), amax(wp.data(), wp.size()));
    if (hipStreamSynchronize(net->stream) != hipSuccess) return azx_net_fail(AZX_EHIP, "net: stream sync before the weight upload failed");
    bool ok = true;
    auto put = [&](const char *name, const void *src, size_t bytes) {
        void *dst = const_cast<void *>(pack_ptr(net, name));
        if (!dst) return;                                 // this network's kernels do not read that operand
        ok = ok && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
#define PUTV(name, vec) put(name, (vec).data(), (vec).size() * sizeof((vec)[0]))
    PUTV("stemT", stemT); PUTV("stem_b", stem_b); PUTV("bias", bias);
    PUTV("Wg", Wg); PUTV("Wp", Wp); PUTV("Ws", Ws); PUTV("Ws16", Ws16); PUTV("Wh16", Wh16);
    PUTV("Whd16", Whd16); PUTV("hbias16", hbias16);
    PUTV("wv", wv); PUTV("bv", bv); PUTV("wp", wp); PUTV("bp", bp);
    PUTV("fc2T", fc2T); PUTV("fc2b", *fc2b); PUTV("fc3w", *fc3w); PUTV("fc3b", *fc3b);
    PUTV("mfcT", mfcT); PUTV("mfcb", mfcb); PUTV("hmP", hmP); PUTV("hmV", hmV);
#undef PUTV
    if (!ok) return azx_net_fail(AZX_EHIP, "net: uploading the packed weights failed");
    (void)plan;
    return AZX_OK;
}

// =================================================================================================================
// device pack: the same arithmetic, one thread per packed element (or per 8-element fragment row)
// =================================================================================================================
// No FMA contraction anywhere in the folding arithmetic: the host reference runs on baseline x86-64 (no FMA), and
// the two must agree bit for bit.
#pragma clang fp contract(off)

template <class F>
__global__ __launch_bounds__(256) void k_each(size_t n, F f) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f(i, i < n);
}
template <class F>
static void each(hipStream_t st, size_t n, F f) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_each<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, f);
}

__device__ __forceinline__ void wave_atomic_max(uint32_t *dst, float a) {
    uint32_t b = __float_as_uint(fabsf(a));      // NaN orders above every finite value and above inf
    for (int o = 32; o >= 1; o >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0) atomicMax(dst, b);
}
__device__ __forceinline__ float fmax_abs(float m, float v) {
    const float a = fabsf(v);
    return !(a <= m) ? a : m;                    // keeps a NaN once seen
}
__device__ __forceinline__ void split_bits(float w, unsigned short &hi, unsigned short &lo) {
    const _Float16 h = (_Float16)w;
    const _Float16 l = (_Float16)(w - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = (unsigned short)((__builtin_bit_cast(unsigned short, l) + LO_RND) & LO_MASK);
}
__device__ __forceinline__ float folded(const float *w, const double *sc, int C, int co, int ci, int tap) {
    return (float)((double)w[((size_t)co * C + ci) * 9 + tap] * sc[co]);
}
struct Frag8 { unsigned short h[8], l[8]; };
__device__ __forceinline__ void store_frag(unsigned short *dst_hi, unsigned short *dst_lo, const Frag8 &f) {
    uint4 a, b;
    a.x = f.h[0] | ((uint32_t)f.h[1] << 16); a.y = f.h[2] | ((uint32_t)f.h[3] << 16);
    a.z = f.h[4] | ((uint32_t)f.h[5] << 16); a.w = f.h[6] | ((uint32_t)f.h[7] << 16);
    b.x = f.l[0] | ((uint32_t)f.l[1] << 16); b.y = f.l[2] | ((uint32_t)f.l[3] << 16);
    b.z = f.l[4] | ((uint32_t)f.l[5] << 16); b.w = f.l[6] | ((uint32_t)f.l[7] << 16);
    *reinterpret_cast<uint4 *>(dst_hi) = a;
    *reinterpret_cast<uint4 *>(dst_lo) = b;
}

static int pack_device(AzxNet *net, const Plan &p) {
    hipStream_t st = net->stream;
    NetDev &d = net->d;
    const float *const *tab = net->raw_tab;
    const int C = p.C, n2 = p.n2, L = p.L, nf = C * (L + 1) + 6;
    double *sc = net->fold, *sh = net->fold + nf;
    uint32_t *wmax = net->wmax;
    if (hipMemsetAsync(wmax, 0, sizeof(uint32_t) * (L + 2), st) != hipSuccess)
        return azx_net_fail(AZX_EHIP, "net: clearing the weight range words failed");
    float *stem_b = const_cast<float *>(d.stem_b), *bias = const_cast<float *>(d.bias), *bv = const_cast<float *>(d.bv),
          *bp = const_cast<float *>(d.bp), *hbias16 = const_cast<float *>(d.hbias16);

    // 1. eval-mode BatchNorm -> scale / shift per channel (network.py:21, :48; eps 1e-5), f64
    each(st, (size_t)nf, [=] __device__(size_t i, bool on) {
        if (!on) return;
        const int f = (int)i;
        int base, c;
        if (f < C) { base = RT_BN1; c = f; }
        else if (f < C * (L + 1)) { const int l = f / C - 1; base = RT_LAYER0 + 5 * l + 1; c = f - C * (l + 1); }
        else if (f < C * (L + 1) + 2) { base = RT_VBN; c = f - C * (L + 1); }
        else { base = RT_PBN; c = f - C * (L + 1) - 2; }
        const double w = tab[base][c], b = tab[base + 1][c], m = tab[base + 2][c], v = tab[base + 3][c];
        const double scale = w / sqrt(v + 1e-5);
        const double shift = b - m * scale;
        sc[f] = scale;
        sh[f] = shift;
        if (f < C) stem_b[c] = (float)shift;
        else if (f < C * (L + 1)) bias[f - C] = (float)shift;
        else if (f < C * (L + 1) + 2) { bv[c] = (float)shift; if (hbias16) hbias16[c] = (float)shift; }
        else { bp[c] = (float)shift; if (hbias16) hbias16[2 + c] = (float)shift; }
    });

    // 2. stem table T[tap][v][co] = scale[co] * sum_i emb[v][i] * w[co][i][tap]; row 27 stays zero
    float *stemT = const_cast<float *>(d.stemT);
    each(st, (size_t)27 * C, [=] __device__(size_t i, bool on) {
        float out = 0.f;
        if (on) {
            const int k = (int)(i / C), co = (int)(i % C), tap = k / 3, v = k % 3;
            const float *emb = tab[RT_EMB], *w1 = tab[RT_CONV1];
            double s = 0;
            for (int j = 0; j < 4; ++j) s += (double)emb[v * 4 + j] * (double)w1[(co * 4 + j) * 9 + tap];
            out = (float)(s * sc[co]);
            stemT[i] = out;
        }
        wave_atomic_max(wmax, out);
    });
    if (p.s32) {
        unsigned short *Ws = const_cast<unsigned short *>(d.Ws);
        const int NT = p.NT;
        each(st, (size_t)2 * NT * 64, [=] __device__(size_t i, bool on) {     // [kk][ntile][hi,lo][lane][t]
            if (!on) return;
            const int ln = (int)(i % 64), nt = (int)(i / 64 % NT), kk = (int)(i / 64 / NT);
            const int j = ln & 31, h = ln >> 5, co = 32 * nt + j;
            Frag8 f;
            for (int t = 0; t < 8; ++t) {
                const int k = 16 * kk + 8 * h + t;
                split_bits(k < 27 ? stemT[(size_t)k * C + co] : 0.0f, f.h[t], f.l[t]);
            }
            unsigned short *base = Ws + ((size_t)(kk * NT + nt) * 2) * 64 * 8;
            store_frag(base + (size_t)ln * 8, base + (size_t)(64 + ln) * 8, f);
        });
    }
    if (p.s16) {
        unsigned short *Ws16 = const_cast<unsigned short *>(d.Ws16);
        each(st, (size_t)p.NT16 * 64, [=] __device__(size_t i, bool on) {     // [ntile][hi,lo][lane][t]
            if (!on) return;
            const int ln = (int)(i % 64), nt = (int)(i / 64);
            const int j = ln & 15, h = ln >> 4, co = 16 * nt + j;
            Frag8 f;
            for (int t = 0; t < 8; ++t) {
                const int k = 8 * h + t;
                split_bits(k < 27 ? stemT[(size_t)k * C + co] : 0.0f, f.h[t], f.l[t]);
            }
            unsigned short *base = Ws16 + ((size_t)nt * 2) * 64 * 8;
            store_frag(base + (size_t)ln * 8, base + (size_t)(64 + ln) * 8, f);
        });
    }

    // 3. tower filters, BN scale folded in, in the order the launched kernels walk them
    if (p.wg) {
        float *Wg = const_cast<float *>(d.Wg);
        each(st, p.n_Wg, [=] __device__(size_t i, bool on) {                   // [layer][tap][cin][cout]
            float v = 0.f;
            int l = 0;
            if (on) {
                const int co = (int)(i % C), ci = (int)(i / C % C), tap = (int)(i / C / C % 9);
                l = (int)(i / C / C / 9);
                v = folded(tab[RT_LAYER0 + 5 * l], sc + C * (l + 1), C, co, ci, tap);
                Wg[i] = v;
            }
            wave_atomic_max(wmax + 1 + __shfl(l, 0), v);      // 9 C^2 is a multiple of 64: one layer per wave
        });
    }
    if (p.wp) {
        float *Wp = const_cast<float *>(d.Wp);
        const int NT = p.NT, Q = C / 8;
        each(st, p.n_Wp / 4, [=] __device__(size_t i, bool on) {               // [layer][tap][q][ntile][lane][t]
            float m = 0.f;
            int l = 0;
            if (on) {
                const int ln = (int)(i % 64), nt = (int)(i / 64 % NT), q = (int)(i / 64 / NT % Q), tap = (int)(i / 64 / NT / Q % 9);
                l = (int)(i / 64 / NT / Q / 9);
                const int j = ln & 31, h = ln >> 5, co = 32 * nt + j;
                float4 o;
                float *po = &o.x;
                for (int t = 0; t < 4; ++t) {
                    po[t] = folded(tab[RT_LAYER0 + 5 * l], sc + C * (l + 1), C, co, 8 * q + 4 * h + t, tap);
                    m = fmax_abs(m, po[t]);
                }
                reinterpret_cast<float4 *>(Wp)[i] = o;
            }
            wave_atomic_max(wmax + 1 + __shfl(l, 0), m);
        });
    }
    if (p.s16 && L > 0) {
        unsigned short *Wh16 = const_cast<unsigned short *>(d.Wh16);
        const int NT16 = p.NT16, NCH = p.NCH;
        const bool wide = net->tower_variant == 5;
        // [layer][tap][chunk][half][ntile][hi,lo][lane][t]
        each(st, (size_t)L * 9 * NCH * 2 * NT16 * 64, [=] __device__(size_t i, bool on) {
            float m = 0.f;
            int l = 0;
            if (on) {
                size_t r = i;
                const int ln = (int)(r % 64); r /= 64;
                const int nt = (int)(r % NT16); r /= NT16;
                const int half = (int)(r % 2); r /= 2;
                const int ch = (int)(r % NCH); r /= NCH;
                const int tap = (int)(r % 9);
                l = (int)(r / 9);
                const int j = ln & 15, h = ln >> 4;
                int co = 16 * nt + j;
                // wide tower: row j = 4 lh + r of tile n = nt % 4 of a wave's 64-channel group is channel
                // 32 (n >> 1) + 8 lh + 4 (n & 1) + r of the group (k_conv_wide_f16x3_s16's 16-byte epilogue pieces)
                if (wide) co = 64 * (nt / 4) + 32 * ((nt % 4) >> 1) + 8 * (j >> 2) + 4 * (nt & 1) + (j & 3);
                Frag8 f;
                for (int t = 0; t < 8; ++t) {
                    const float v = folded(tab[RT_LAYER0 + 5 * l], sc + C * (l + 1), C, co, 64 * ch + 32 * half + 8 * h + t, tap);
                    m = fmax_abs(m, v);
                    split_bits(v, f.h[t], f.l[t]);
                }
                unsigned short *base = Wh16 + (i / 64) * 2 * 64 * 8;
                store_frag(base + (size_t)ln * 8, base + (size_t)(64 + ln) * 8, f);
            }
            wave_atomic_max(wmax + 1 + __shfl(l, 0), m);
        });
    }

    // 4. heads: the two 1x1 convs with their BN folded in, the FC layers transposed / tiled
    {
        float *wv = const_cast<float *>(d.wv), *wp = const_cast<float *>(d.wp);
        const int fv = C * (L + 1), fp = fv + 2;
        each(st, (size_t)6 * C, [=] __device__(size_t i, bool on) {
            float v = 0.f;
            if (on) {
                const int o = (int)(i / C), c = (int)(i % C);
                if (o < 2) { v = (float)((double)tab[RT_VCONV][(size_t)o * C + c] * sc[fv + o]); wv[(size_t)o * C + c] = v; }
                else { v = (float)((double)tab[RT_PCONV][(size_t)(o - 2) * C + c] * sc[fp + o - 2]); wp[(size_t)(o - 2) * C + c] = v; }
            }
            wave_atomic_max(wmax + 1 + L, v);
        });
        if (p.hd16) {
            unsigned short *Whd16 = const_cast<unsigned short *>(d.Whd16);
            each(st, (size_t)2 * 64, [=] __device__(size_t i, bool on) {        // [kstep][hi,lo][lane][t]
                if (!on) return;
                const int ln = (int)(i % 64), ks = (int)(i / 64), o = ln & 15;
                Frag8 f;
                for (int t = 0; t < 8; ++t) {
                    const int ci = 32 * ks + 8 * (ln >> 4) + t;
                    const float w = o < 2 ? wv[(size_t)o * C + ci] : o < 6 ? wp[(size_t)(o - 2) * C + ci] : 0.0f;
                    split_bits(w, f.h[t], f.l[t]);
                }
                unsigned short *base = Whd16 + (size_t)ks * 2 * 64 * 8;
                store_frag(base + (size_t)ln * 8, base + (size_t)(64 + ln) * 8, f);
            });
        }
        float *fc2T = const_cast<float *>(d.fc2T), *fc2b = const_cast<float *>(d.fc2b), *fc3w = const_cast<float *>(d.fc3w),
              *fc3b = const_cast<float *>(d.fc3b), *mfcT = const_cast<float *>(d.mfcT), *mfcb = const_cast<float *>(d.mfcb),
              *hmP = const_cast<float *>(d.hmP), *hmV = const_cast<float *>(d.hmV);
        each(st, p.n_fc2T, [=] __device__(size_t i, bool on) {                  // fc2T[i][o] = value_fc2.weight[o][i]
            if (!on) return;
            const int o = (int)(i % 64), k = (int)(i / 64);
            fc2T[i] = tab[RT_FC2W][(size_t)o * 2 * n2 + k];
            if (i < 64) { fc2b[i] = tab[RT_FC2B][i]; fc3w[i] = tab[RT_FC3W][i]; }
            if (i == 0) fc3b[0] = tab[RT_FC3B][0];
        });
        each(st, (size_t)4 * n2 * n2, [=] __device__(size_t i, bool on) {       // mfcT[i][t] = move_fc.weight[t][i]
            if (!on) return;
            const int t = (int)(i % n2), k = (int)(i / n2);
            mfcT[(size_t)k * AZX_CELL_STRIDE + t] = tab[RT_MFW][(size_t)t * 4 * n2 + k];
            if (i < (size_t)n2) mfcb[i] = tab[RT_MFB][i];
        });
        // k_heads_mfma's B operands: [n tile of 16][k group of 16][lane][s] = W[k = 16 q + 4 (lane >> 4) + s][unit 16 tile + (lane & 15)]
        const int QP = p.KPp / 16, QV = p.KVp / 16;
        each(st, p.n_hmP, [=] __device__(size_t i, bool on) {
            if (!on) return;
            const int s = (int)(i % 4), l = (int)(i / 4 % 64), q = (int)(i / 256 % QP), t = (int)(i / 256 / QP);
            const int k = 16 * q + 4 * (l >> 4) + s, unit = 16 * t + (l & 15);
            hmP[i] = (k < 4 * n2 && unit < n2) ? tab[RT_MFW][(size_t)unit * 4 * n2 + k] : 0.f;
        });
        each(st, p.n_hmV, [=] __device__(size_t i, bool on) {
            if (!on) return;
            const int s = (int)(i % 4), l = (int)(i / 4 % 64), q = (int)(i / 256 % QV), t = (int)(i / 256 / QV);
            const int k = 16 * q + 4 * (l >> 4) + s, unit = 16 * t + (l & 15);
            hmV[i] = (k < 2 * n2) ? tab[RT_FC2W][(size_t)unit * 2 * n2 + k] : 0.f;
        });
    }
    if (hipGetLastError() != hipSuccess) return azx_net_fail(AZX_EHIP, "net: a weight pack kernel failed to launch");
    if (hipMemcpyAsync(net->wmax_host, wmax, sizeof(uint32_t) * (L + 2), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return azx_net_fail(AZX_EHIP, "net: the weight pack did not complete");
    return AZX_OK;
}

// =================================================================================================================
// entry points
// =================================================================================================================
int azx_net_set_weights(AzxNet *net, int n, const char *const *names, const void *const *ptrs,
                        const int64_t *counts, int on_device) {
    const Plan plan = make_plan(net);
    const std::vector<RawSlot> slots = raw_slots(plan.C, plan.n2, plan.L);
    if (int rc = ensure_buffers(net, plan)) return rc;
    // every tensor of the state_dict this network needs, by name, with its size
    std::map<std::string, int> given;
    for (int i = 0; i < n; ++i) given[names[i]] = i;
    std::vector<int> src(slots.size());
    for (size_t s = 0; s < slots.size(); ++s) {
        auto it = given.find(slots[s].name);
        if (it == given.end() || (size_t)counts[it->second] != slots[s].count || !ptrs[it->second]) {
            const std::string msg = "net: tensor '" + slots[s].name + "' missing or has the wrong size";
            return azx_net_fail(AZX_EINVAL, msg.c_str());
        }
        src[s] = it->second;
    }
    const int L = plan.L;
    std::vector<float> wmax(L + 2, 0.f);
    net->ready = false;                       // a rejected or failed update leaves the engine without weights
    if (net->pack_on_host) {
        std::map<std::string, std::vector<float>> T;
        for (size_t s = 0; s < slots.size(); ++s) {
            std::vector<float> h(slots[s].count);
            if (on_device) {
                if (hipMemcpy(h.data(), ptrs[src[s]], h.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
                    return azx_net_fail(AZX_EHIP, "net: copying a weight tensor from the device failed");
            } else {
                memcpy(h.data(), ptrs[src[s]], h.size() * sizeof(float));
            }
            T[slots[s].name] = std::move(h);
        }
        if (int rc = pack_host(net, plan, T, wmax.data())) return rc;
    } else {
        if (!on_device) {
            // host arrays: staged through one device arena (allocated on first use), then packed like live tensors
            size_t total = 0;
            for (const RawSlot &s : slots) total += (s.count + 3) & ~(size_t)3;
            if (!net->raw_arena) {
                net->raw_arena = nalloc<float>(net, total);
                net->raw_arena_floats = total;
                if (!net->raw_arena) return azx_net_fail(AZX_ENOMEM, "net: allocating the weight staging arena failed");
                if (hipDeviceSynchronize() != hipSuccess) return azx_net_fail(AZX_EHIP, "net: device sync failed");   // its zero fill (null stream)
            }
            // queued kernels of an earlier pack may still read the arena
            if (hipStreamSynchronize(net->stream) != hipSuccess) return azx_net_fail(AZX_EHIP, "net: stream sync failed");
            size_t o = 0;
            for (size_t s = 0; s < slots.size(); ++s) {
                if (hipMemcpyAsync(net->raw_arena + o, ptrs[src[s]], slots[s].count * sizeof(float), hipMemcpyHostToDevice, net->stream) != hipSuccess)
                    return azx_net_fail(AZX_EHIP, "net: copying a weight tensor to the device failed");
                net->raw_tab_host[s] = net->raw_arena + o;
                o += (slots[s].count + 3) & ~(size_t)3;
            }
        } else {
            for (size_t s = 0; s < slots.size(); ++s) net->raw_tab_host[s] = static_cast<const float *>(ptrs[src[s]]);
        }
        if (hipMemcpyAsync(net->raw_tab, net->raw_tab_host, sizeof(const float *) * slots.size(), hipMemcpyHostToDevice, net->stream) != hipSuccess)
            return azx_net_fail(AZX_EHIP, "net: uploading the tensor table failed");
        if (int rc = pack_device(net, plan)) return rc;
        for (int g = 0; g < L + 2; ++g) memcpy(&wmax[g], &net->wmax_host[g], 4);
    }
    // range guard: the split-f16 towers carry every folded weight as hi + lo f16 halves
    const bool f16 = net->tower_variant == 4 || net->tower_variant == 5;
    for (int g = 0; g < L + 2; ++g) {
        const bool finite = wmax[g] == wmax[g] && wmax[g] < INFINITY;
        if (finite && (!f16 || wmax[g] <= F16_MAX)) continue;
        if (g == L + 1 && finite && !plan.hd16) continue;       // the head convs are only split when the fused tower runs them
        char what[160];
        if (g == 0) snprintf(what, sizeof what, "conv1.weight folded with bn1 and the embedding");
        else if (g <= L) snprintf(what, sizeof what, "resblocks.%d.conv%d.weight folded with resblocks.%d.bn%d", (g - 1) / 2, (g - 1) % 2 + 1, (g - 1) / 2, (g - 1) % 2 + 1);
        else snprintf(what, sizeof what, "value_conv1 / move_conv1 weights folded with their BatchNorm");
        char msg[400];
        if (!finite) snprintf(msg, sizeof msg, "net: %s is not finite", what);
        else snprintf(msg, sizeof msg, "net: %s reaches |w| = %.6g, beyond the f16 range (65504) of the split-f16 tower; "
                      "the weights were NOT installed (AZX_TOWER=fp32 runs the fp32 MFMA tower instead)", what, (double)wmax[g]);
        return azx_net_fail(AZX_ERANGE, msg);
    }
    if (net->d.sat_flag && hipMemsetAsync(net->d.sat_flag, 0, sizeof(uint32_t), net->stream) != hipSuccess)
        return azx_net_fail(AZX_EHIP, "net: clearing the activation range flag failed");
    net->ready = true;
    return AZX_OK;
}

int azx_net_debug_weights(AzxNet *net, int which, void *out, int64_t cap, int64_t *nbytes, char *name, int name_cap) {
    if (!net || which < 0 || !nbytes) return azx_net_fail(AZX_EINVAL, "net: bad argument");
    if (!net->ready) return azx_net_fail(AZX_ESTATE, "net: azx_set_weights has not been called");
    if ((size_t)which >= net->packs.size()) { *nbytes = -1; return AZX_OK; }
    const AzxNet::PackBuf &b = net->packs[which];
    *nbytes = (int64_t)b.bytes;
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", b.name.c_str());
    if (out && cap >= (int64_t)b.bytes) {
        if (hipStreamSynchronize(net->stream) != hipSuccess ||
            hipMemcpy(out, b.ptr, b.bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return azx_net_fail(AZX_EHIP, "net: copying a packed buffer to the host failed");
    }
    return AZX_OK;
}
