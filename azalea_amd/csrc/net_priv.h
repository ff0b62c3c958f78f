// net_priv.h -- what net_kernels.hip (the forward kernels) and net_pack.hip (the weight packers) share:
// the packed-weight table the kernels read (NetDev), the host-side network object (AzxNet) and the
// hi/lo f16 split's rounding of the `lo` half.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <string>
#include <vector>

#include "net.h"

struct NetDev {
    int N, ncells, C, blocks, layers;     // layers = 2*blocks
    // stem: embedding folded through conv1+bn1 (network.py:125,:141-142,:47-48,:73)
    const float *stemT;    // [9][3][C]   table[tap][cell value][cout], then one all-zero row
    const float *stem_b;   // [C]
    const unsigned short *Ws;  // wide stem: f16x3 pack of stemT as a 32x32x16 MFMA A operand: [2 kk][ntile][hi,lo][64 lanes][8] f16 bits
    const unsigned short *Ws16, *Wh16;   // the same weights in 16x16x32 fragment order (k_tower_f16x3_s16)
    const unsigned short *Whd16;         // the heads' six 1x1 conv filters as one 16-row A tile: [kstep 2][hi,lo][lane][8]
    const float *hbias16;                // their folded-BN biases, padded to 16
    // tower (network.py:17-39, :50-52): BN folded into the conv weights
    const float *Wp;       // MFMA pack [layers][9][C/8][C/32][64][4]
    const float *Wg;       // generic   [layers][9][C][C]  (tap, cin, cout)
    const float *bias;     // [layers][C]
    // heads (network.py:54-60, :77-84, :127-128, :146)
    const float *wv, *bv;  // [2][C], [2]
    const float *wp, *bp;  // [4][C], [4]
    const float *fc2T, *fc2b;   // [2*ncells][64], [64]
    const float *fc3w, *fc3b;   // [64], [1]
    const float *mfcT, *mfcb;   // [4*ncells][AZX_CELL_STRIDE], [AZX_CELL_STRIDE]
    // the two FC weight matrices as fp32-MFMA B operands (k_heads_mfma): [n tile][k group of 8][64 lanes][4]
    const float *hmP, *hmV;
    int hm_lda;                 // LDS row stride (floats) of k_heads_mfma's feature tile
    // range guard of the split-f16 towers: an activation above the f16 range (65 504) would become +inf in its `hi`
    // half.  The epilogues keep the largest `hi` they stored and OR 1 in here when it is not finite; the host reads
    // the word where it synchronises anyway and reports AZX_ERANGE (azx_net_check_range).
    uint32_t *sat_flag;
};

// The `lo` halves of the hi+lo f16 split (activations: split2_f16; weights: the packers) keep AZX_LO_BITS explicit
// mantissa bits, rounded to nearest (10 = all of them; DESIGN 3.2 has the precision / clock trade measured).
#ifndef AZX_LO_BITS
#define AZX_LO_BITS 10
#endif
#define LO_MASK ((0xFFFFu << (10 - AZX_LO_BITS)) & 0xFFFFu)
#define LO_RND (AZX_LO_BITS < 10 ? (1u << (9 - AZX_LO_BITS)) : 0u)
static inline unsigned short lo_round_bits(unsigned short bits) {
    return (unsigned short)((bits + LO_RND) & LO_MASK);
}

struct AzxNet {
    NetDev d;
    int max_evals = 0;
    bool ready = false;
    bool use_mfma = false;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    float *act = nullptr, *act2 = nullptr, *act3 = nullptr;   // [E][ncells][C]
    unsigned short *wideX = nullptr, *wideY = nullptr;        // wide tower: [E][ncells][C hi | C lo] f16
    float *logit = nullptr;                                     // [E][AZX_CELL_STRIDE]
    float *hfeat = nullptr;                                     // [E][6][ncells] head features from the fused tower
    // host-forward staging
    uint8_t *hb_board = nullptr;
    int32_t *hb_flip = nullptr;
    float *hb_value = nullptr;
    size_t lds_bytes = 0;
    int tower_variant = 0;
    // wide tower: the second half of a batch's boards runs its layer launches on a second stream
    hipStream_t stream2[3] = {nullptr, nullptr, nullptr};
    std::vector<uint32_t> cu_mask;   // azx_net_set_stream: the side streams are made on the engine's CU mask (empty = all CUs)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    bool streams_ok = false;
    // diagnostic switches, read once per engine by azx_net_create (azx_net_kernel_info reports the outcome)
    int opt_wsplit = 2;         // AZX_WIDE_STREAMS: streams the wide tower's layer launches are spread over
    bool opt_heads_mfma = true; // AZX_HEADS=valu: the scalar-FMA k_heads behind the fused tower too
    std::string info;
    // ---- weight packing (net_pack.hip) ----
    bool pack_on_host = false;       // AZX_PACK=host: the host reference pack (debug / bit-identity tests)
    bool packed_once = false;        // the persistent packed buffers exist
    const float **raw_tab = nullptr; // device table of the raw state_dict tensors, fixed slot order (net_pack.hip)
    const float **raw_tab_host = nullptr;   // its pinned host staging
    int raw_slots = 0;
    float *raw_arena = nullptr;      // device staging of host-memory tensors (on_device == 0)
    size_t raw_arena_floats = 0;
    double *fold = nullptr;          // [2][fold_channels] folded-BN scale | shift (f64, like the host pack)
    uint32_t *wmax = nullptr;        // [layers + 2] largest |folded weight| bits per tensor group (stem, convs, head convs)
    uint32_t *wmax_host = nullptr;   // pinned
    struct PackBuf { std::string name; void *ptr; size_t bytes; };
    std::vector<PackBuf> packs;      // every packed buffer by name (azx_net_debug_weights, digests)
};

int azx_net_fail(int code, const char *msg);     // sets azx_net_error()'s text, returns code

template <typename T>
static T *nalloc(AzxNet *net, size_t count) {
    void *p = nullptr;
    if (hipMalloc(&p, std::max<size_t>(count * sizeof(T), 16)) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, std::max<size_t>(count * sizeof(T), 16));
    net->allocs.push_back(p);
    return reinterpret_cast<T *>(p);
}
