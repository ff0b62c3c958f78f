// azx_capi.cpp -- host side of the C ABI declared in include/azx.h.
// Owns the device arenas, sequences the kernels on one HIP stream per engine, and converts
// between the reference's data model (six tree arrays, ragged legal-move lists) and the
// engine's HBM layout.  No CPU fallback: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/azx.h"
#include "azx_dev.h"
#include "mcts_kernels.h"
#include "net.h"
#include "train.h"
#include "replay_kernels.h"

#ifndef AZX_SRC_SHA
#define AZX_SRC_SHA "unknown"      // the Makefile passes the digest of the kernel sources (profiles are keyed to it)
#endif

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHECK(expr)                                                                    \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            return fail(AZX_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                        __FILE__, __LINE__);                                              \
    } while (0)

struct azx_engine {
    azx_config cfg;
    DevEngine d;
    hipStream_t stream = nullptr;
    int reserved_cus = 0;               // azx_reserve_cus: CUs of the device the engine's streams stay off
    std::mutex alloc_mu;                // dev_alloc from the trainer's thread (collate staging) beside the play thread's
    int num_batches = 0;
    int selects_per_search = 0;
    std::vector<void *> allocs;
    // gather buffers
    int32_t *g_k = nullptr, *g_legal = nullptr, *g_nn = nullptr;
    float *g_cv = nullptr, *g_cw = nullptr, *g_cp = nullptr, *g_rv = nullptr, *g_rw = nullptr,
          *g_sv = nullptr;
    double *noise_dev = nullptr;
    size_t noise_cap = 0;
    float *prior_table_dev = nullptr;
    int32_t *moveids_dev = nullptr;
    int32_t *slots_dev = nullptr, *moves_dev = nullptr, *nmoves_dev = nullptr;
    size_t moves_cap = 0;
    // external evaluator bookkeeping
    bool ext_active = false;
    int ext_batches_done = 0;
    std::vector<int> ext_order;                 // eval indices sorted by (slot, leaf)
    std::vector<std::vector<int>> ext_cells;    // original legal cells per sorted entry
    // play mode
    bool play_ready = false;
    int64_t q_alloc = 0;
    int64_t q_rows_valid = 0;           // rows the last azx_play_device left in the queue
    std::vector<void *> q_allocs;
    int32_t *export_board = nullptr;    // azx_play's device-side widening staging
    float *export_prob = nullptr;
    size_t export_cap = 0;
    // device-resident replay ring (azx_replay_*)
    ReplayRows ring = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t ring_cap = 0, ring_size = 0, ring_write = 0;
    long long *ring_idx = nullptr;      // sampled row indices of the collate in flight
    int64_t ring_idx_cap = 0;
    int32_t *ring_maxk = nullptr;
    bool ring_mover_view = false;       // azx_replay_set_mover_view
    std::vector<void *> ring_allocs;
    // azx_replay_collate_async: index staging, AZX_COLLATE_SLOTS deep (pinned host + device), one event per slot
    long long *cidx_host[8] = {nullptr}, *cidx_dev[8] = {nullptr};
    hipEvent_t cidx_ev[8] = {nullptr};
    int64_t cidx_cap = 0;
    uint64_t cidx_next = 0;
    int32_t *cidx_maxk = nullptr;
    // timing
    std::vector<hipEvent_t> ev_pool;
    std::vector<char> ev_tag;           // 0 = tree kernel, 1 = network (tower + heads)
    std::vector<int> ev_weight;         // moves covered by the timed launch (k_play: several)
    size_t ev_used = 0;
    AzxNet *net = nullptr;
    // diagnostic switches, read once at azx_create (azx_kernel_info reports them)
    bool force_generic = false;         // AZX_MCTS_GENERIC: every tree launch on the generic instantiation
    bool no_persistent = false;         // AZX_NO_PERSISTENT: per-move launches instead of k_play
    int64_t dbg_qcap = 0;               // azx_debug_set_queue_cap
};

template <typename T>
static int dev_alloc(azx_engine *e, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t err = hipMalloc(&q, bytes);
    if (err != hipSuccess)
        return fail(AZX_ENOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
    if (zero) {
        err = hipMemsetAsync(q, 0, bytes, e->stream);
        if (err != hipSuccess) return fail(AZX_EHIP, "hipMemset failed: %s", hipGetErrorString(err));
    }
    std::lock_guard<std::mutex> lock(e->alloc_mu);
    e->allocs.push_back(q);
    *p = reinterpret_cast<T *>(q);
    return AZX_OK;
}

// Every entry point runs with the engine's device current and puts the caller's device back on
// return (hipSetDevice is per host thread and also moves torch.cuda.current_device()).
struct DevGuard {
    int prev = -1;
    bool changed = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() { if (changed) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
};
#define ENGINE_GUARD(e) DevGuard _dev_guard((e)->cfg.device)

#define TRY(expr)            \
    do {                     \
        int _rc = (expr);    \
        if (_rc) return _rc; \
    } while (0)

extern "C" const char *azx_last_error(void) { return g_err.c_str(); }
extern "C" int azx_version(void) { return 5; }   // 5: AZX_ERANGE, azx_debug_weights, device-side weight pack

extern "C" int azx_create(const azx_config *cfg, azx_engine **out) {
    if (!cfg || !out) return fail(AZX_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AZX_ENODEV, "no HIP device visible: the engine has no CPU fallback");
    if (cfg->board_size < 2 || cfg->board_size > AZX_MAX_BOARD)
        return fail(AZX_EINVAL, "board_size %d outside [2, %d]", cfg->board_size, AZX_MAX_BOARD);
    if (cfg->n_games < 1) return fail(AZX_EINVAL, "n_games must be >= 1");
    if (cfg->search_batch_size < 1 || cfg->search_batch_size > AZX_MAX_BATCH)
        return fail(AZX_EINVAL, "search_batch_size %d outside [1, %d]", cfg->search_batch_size,
                    AZX_MAX_BATCH);
    if (cfg->simulations < 0) return fail(AZX_EINVAL, "simulations must be >= 0");
    if (cfg->evaluator < AZX_EVAL_RESNET || cfg->evaluator > AZX_EVAL_EXTERNAL)
        return fail(AZX_EINVAL, "unknown evaluator %d", cfg->evaluator);
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(AZX_EINVAL, "device %d not in [0, %d)", cfg->device, ndev);
    if (cfg->game_index_stride < 0 || cfg->game_index_offset < 0 ||
        cfg->game_index_offset >= std::max(1, cfg->game_index_stride))
        return fail(AZX_EINVAL, "game_index_offset %d outside [0, game_index_stride %d)", cfg->game_index_offset,
                    std::max(1, cfg->game_index_stride));
    DevGuard guard(cfg->device);
    { int cur = -1; if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) return fail(AZX_EHIP, "hipSetDevice(%d) failed", cfg->device); }
    if (azx_init_geometry(cfg->device)) return fail(AZX_EHIP, "uploading the board geometry tables failed");

    azx_engine *e = new azx_engine();
    e->cfg = *cfg;
    { const char *v = getenv("AZX_MCTS_GENERIC"); e->force_generic = v && atoi(v) != 0; }
    { const char *v = getenv("AZX_NO_PERSISTENT"); e->no_persistent = v && atoi(v) != 0; }
    HIPCHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    DevEngine &d = e->d;
    memset(&d, 0, sizeof d);
    d.N = cfg->board_size;
    d.ncells = d.N * d.N;
    d.G = cfg->n_games;
    d.bs = cfg->search_batch_size;
    d.slots = d.ncells <= 128 ? 2 : 3;
    d.c_puct = cfg->exploration_coef;
    d.evaluator = cfg->evaluator;
    d.flags = cfg->flags;
    d.seed = cfg->seed;
    d.uid_stride = std::max(1, cfg->game_index_stride);
    d.uid_offset = cfg->game_index_offset;
    d.noise_alpha = (float)cfg->noise_alpha;
    d.noise_scale = 0.0;
    d.exploration_depth = cfg->exploration_depth;
    d.temperature = (float)cfg->temperature;
    e->num_batches = cfg->simulations / cfg->search_batch_size + 1;   // mcts.py:268
    e->selects_per_search = e->num_batches * cfg->search_batch_size;
    d.selects_per_search = e->selects_per_search;
    int cap = cfg->nodes_per_game;
    // default: six moves' worth of expansions -- a sharply peaked network carries most of its tree
    // from move to move (the reference allows 10M nodes per game, search_tree.py:18)
    if (cap <= 0) cap = 6 * (e->selects_per_search + 1) * d.ncells + 1024;
    d.cap = cap;

    const size_t G = d.G, bs = d.bs, E = G * bs;
    const size_t pstride = d.ncells + (d.ncells & 1);
    int rc = AZX_OK;
#define A(ptr, count) if (!rc) rc = dev_alloc(e, &ptr, (count))
    A(d.cells, G * d.slots * 64);
    A(d.ghdr, G);
    A(d.thdr, G);
    if (!rc) rc = dev_alloc(e, &d.arena[0], G * (size_t)cap, false);
    if (cfg->flags & AZX_FLAG_NO_COMPACT) d.arena[1] = d.arena[0];
    else if (!rc) rc = dev_alloc(e, &d.arena[1], G * (size_t)cap, false);
    A(d.leaf_node, E); A(d.leaf_len, E); A(d.leaf_eval, E); A(d.leaf_link, E); A(d.leaf_cells, E);
    A(d.leaf_mask, E * 4); A(d.path, E * pstride);
    A(d.ev_board, E * AZX_CELL_STRIDE); A(d.ev_src, E); A(d.ev_flip, E);
    A(d.ev_value, E); A(d.ev_prior, E * AZX_CELL_STRIDE); A(d.n_eval, 4);
    A(d.counters, G * CTR_COUNT); A(d.q_count, 2); A(d.stat_sums, G * 8);
    A(e->g_k, G); A(e->g_legal, G * d.ncells); A(e->g_nn, G);
    A(e->g_cv, G * d.ncells); A(e->g_cw, G * d.ncells); A(e->g_cp, G * d.ncells);
    A(e->g_rv, G); A(e->g_rw, G); A(e->g_sv, G);
    A(e->moveids_dev, G); A(e->slots_dev, G); A(e->nmoves_dev, G);
#undef A
    if (rc) { azx_destroy(e); return rc; }
    if (cfg->evaluator == AZX_EVAL_RESNET) {
        rc = azx_net_create(&e->net, d.N, cfg->num_blocks, cfg->base_chans, (int)E, e->stream);
        if (rc) { g_err = azx_net_error(); azx_destroy(e); return rc; }
    }
    {   // inverse-CDF table of the device Dirichlet sampler for this engine's alpha
        float *gt = nullptr;
        rc = dev_alloc(e, &gt, AZX_GAMMA_TAB_FLOATS);
        if (rc) { azx_destroy(e); return rc; }
        std::vector<float> tab(AZX_GAMMA_TAB_FLOATS, 0.0f);
        if (cfg->noise_alpha > 0.0) azx_gamma_table(cfg->noise_alpha, tab.data());
        if (hipMemcpyAsync(gt, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess) {
            azx_destroy(e);
            return fail(AZX_EHIP, "uploading the gamma sampler table failed");
        }
        e->d.gamma_tab = gt;
    }
    *out = e;
    {   // default uniform prior table: float32 1/k, the same bits as the IEEE division on device
        std::vector<float> tab(d.ncells + 1, 0.0f);
        for (int k = 1; k <= d.ncells; ++k) tab[k] = 1.0f / (float)k;
        rc = azx_set_prior_table(e, tab.data(), d.ncells + 1);
        if (rc) { azx_destroy(e); *out = nullptr; return rc; }
        e->d.prior_default = 1;
    }
    // all slots start as fresh games with uids 0..G-1
    azx_launch_reset(d, nullptr, d.G, nullptr, nullptr, 0, 1, e->stream);
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" void azx_destroy(azx_engine *e) {
    if (!e) return;
    ENGINE_GUARD(e);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->net) azx_net_destroy(e->net);
    for (void *p : e->allocs) (void)hipFree(p);
    for (void *p : e->q_allocs) (void)hipFree(p);
    for (void *p : e->ring_allocs) (void)hipFree(p);
    if (e->export_board) (void)hipFree(e->export_board);
    if (e->export_prob) (void)hipFree(e->export_prob);
    for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
    for (int i = 0; i < 8; ++i) {
        if (e->cidx_ev[i]) { (void)hipEventSynchronize(e->cidx_ev[i]); (void)hipEventDestroy(e->cidx_ev[i]); }
        if (e->cidx_host[i]) (void)hipHostFree(e->cidx_host[i]);
        if (e->cidx_dev[i]) (void)hipFree(e->cidx_dev[i]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" void *azx_stream(azx_engine *e) { return e ? (void *)e->stream : nullptr; }

// CU mask layout of gfx950, measured (tools/microbench/cu_mask.hip, profiles/r6_cu_mask_microbench.txt): bit b of the
// mask is XCD b % 8, shader engine (b / 8) % 4, so bits [0, 8 r) are r CUs of every XCD, dealt one per shader engine.
extern "C" int azx_reserve_cus(azx_engine *e, int cus_per_xcd, int *reserved_out) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, e->cfg.device));
    const int ncu = prop.multiProcessorCount, xcds = 8, ses = 4;
    if (cus_per_xcd < 0 || cus_per_xcd * xcds >= ncu)
        return fail(AZX_EINVAL, "cus_per_xcd %d outside [0, %d)", cus_per_xcd, ncu / xcds);
    if (cus_per_xcd > 0 && ncu % (xcds * ses))
        return fail(AZX_ESTATE, "CU reservation is laid out for 8 XCDs x 4 shader engines; this device has %d CUs", ncu);
    const int per = (cus_per_xcd + ses - 1) / ses * ses;
    const int words = (ncu + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int b = 0; b < ncu; ++b)
        if (b >= per * xcds) mask[b / 32] |= 1u << (b % 32);
    HIPCHECK(hipStreamSynchronize(e->stream));
    hipStream_t fresh = nullptr;
    if (per == 0) HIPCHECK(hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    else HIPCHECK(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)words, mask.data()));
    hipStream_t old = e->stream;
    e->stream = fresh;
    if (e->net) azx_net_set_stream(e->net, fresh, per ? mask.data() : nullptr, per ? words : 0);
    (void)hipStreamDestroy(old);
    e->reserved_cus = per * xcds;
    if (!e->cidx_maxk) TRY(dev_alloc(e, &e->cidx_maxk, 4));     // (so that the trainer's thread never allocates beside a play)
    HIPCHECK(hipStreamSynchronize(e->stream));
    if (reserved_out) *reserved_out = e->reserved_cus;
    return AZX_OK;
}

extern "C" int azx_kernel_info(azx_engine *e, char *buf, int cap) {
    if (!e || !buf || cap < 1) return fail(AZX_EINVAL, "null argument");
    const DevEngine &d = e->d;
    const int S = d.slots <= 2 ? 2 : 3;
    char tree[96], play[96];
    if (d.evaluator == AZX_EVAL_UNIFORM || d.evaluator == AZX_EVAL_UNIFORM_HASH) {
        // the FAST instantiation also needs the default prior table and device (or no) noise: decided per launch
        DevEngine probe = d;
        probe.device_noise = 1;
        const bool fast = azx_mcts_fast_path(probe, MODE_BEGIN | MODE_INLINE, e->force_generic);
        snprintf(tree, sizeof tree, "k_mcts<%d,%s>", S, fast ? "FAST (throughput mode; generic with host noise)" : "generic");
        snprintf(play, sizeof play, "%s", fast && !e->no_persistent ? (S == 2 ? "k_play<2> (persistent)" : "k_play<3> (persistent)")
                                                                       : "k_mcts + k_choose + k_advance per move");
    } else {
        snprintf(tree, sizeof tree, "k_mcts<%d,generic> (BEGIN / APPLY|SELECT / APPLY phases)", S);
        snprintf(play, sizeof play, "phases + k_choose + k_advance per move");
    }
    std::string text = std::string("tree=") + tree + "; play=" + play + "; net=" +
                       (e->net ? azx_net_kernel_info(e->net) : "none") +
                       "; switches: AZX_MCTS_GENERIC=" + (e->force_generic ? "1" : "0") +
                       " AZX_NO_PERSISTENT=" + (e->no_persistent ? "1" : "0") +
                       " reserved_cus=" + std::to_string(e->reserved_cus) +
                       "; src=" AZX_SRC_SHA;       // sha256 (16 hex digits) over the kernel sources this library was built from
    snprintf(buf, (size_t)cap, "%s", text.c_str());
    return (int)text.size();
}

extern "C" int azx_debug_set_queue_cap(azx_engine *e, int64_t rows) {
    if (!e || rows < 0) return fail(AZX_EINVAL, "bad argument");
    e->dbg_qcap = rows;
    return AZX_OK;
}

// AZX_ERANGE when a split-f16 tower launch of this call overflowed an activation (net_kernels.hip: NetDev::sat_flag)
static int check_net_range(azx_engine *e) {
    if (!e->net) return AZX_OK;
    int rc = azx_net_check_range(e->net, e->stream);
    if (rc) g_err = azx_net_error();
    return rc;
}

extern "C" int azx_debug_weights(azx_engine *e, int which, void *out, int64_t cap, int64_t *nbytes, char *name, int name_cap) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (!e->net) return fail(AZX_ESTATE, "engine was not created with AZX_EVAL_RESNET");
    int rc = azx_net_debug_weights(e->net, which, out, cap, nbytes, name, name_cap);
    if (rc) g_err = azx_net_error();
    return rc;
}

extern "C" int azx_set_weights(azx_engine *e, int n_tensors, const char *const *names,
                               const void *const *ptrs, const int64_t *counts, int on_device) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (!e->net) return fail(AZX_ESTATE, "engine was not created with AZX_EVAL_RESNET");
    int rc = azx_net_set_weights(e->net, n_tensors, names, ptrs, counts, on_device);
    if (rc) g_err = azx_net_error();
    return rc;
}

extern "C" int azx_set_prior_table(azx_engine *e, const float *prior_by_k, int count) {
    if (!e || !prior_by_k) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (count < e->d.ncells + 1) return fail(AZX_EINVAL, "prior table needs %d entries", e->d.ncells + 1);
    if (!e->prior_table_dev) TRY(dev_alloc(e, &e->prior_table_dev, (size_t)e->d.ncells + 1));
    HIPCHECK(hipMemcpyAsync(e->prior_table_dev, prior_by_k, sizeof(float) * (e->d.ncells + 1),
                            hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    e->d.prior_by_k = e->prior_table_dev;
    e->d.prior_default = 0;
    return AZX_OK;
}

extern "C" int azx_reset(azx_engine *e, const int32_t *slots, int n_slots, const int32_t *moves,
                         const int32_t *n_moves, int stride) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    if (!slots) n_slots = d.G;
    if (n_slots < 1 || n_slots > d.G) return fail(AZX_EINVAL, "n_slots %d outside [1, %d]", n_slots, d.G);
    if (slots) {
        for (int i = 0; i < n_slots; ++i)
            if (slots[i] < 0 || slots[i] >= d.G) return fail(AZX_EINVAL, "slot %d out of range", slots[i]);
        HIPCHECK(hipMemcpyAsync(e->slots_dev, slots, sizeof(int32_t) * n_slots, hipMemcpyHostToDevice, e->stream));
    }
    if (moves) {
        if (!n_moves || stride < 1) return fail(AZX_EINVAL, "moves given without n_moves/stride");
        const size_t need = (size_t)n_slots * stride;
        if (need > e->moves_cap) {
            TRY(dev_alloc(e, &e->moves_dev, need));
            e->moves_cap = need;
        }
        // validate on the host: the device step assumes legal moves (hex.py:173-176 asserts)
        for (int i = 0; i < n_slots; ++i) {
            if (n_moves[i] < 0 || n_moves[i] > stride) return fail(AZX_EINVAL, "n_moves[%d] out of range", i);
            std::vector<char> used(d.ncells, 0);
            for (int p = 0; p < n_moves[i]; ++p) {
                const int mv = moves[(size_t)i * stride + p];
                if (mv < 1 || mv > d.ncells || used[mv - 1]) return fail(AZX_EINVAL, "illegal move %d", mv);
                used[mv - 1] = 1;
            }
        }
        HIPCHECK(hipMemcpyAsync(e->moves_dev, moves, sizeof(int32_t) * need, hipMemcpyHostToDevice, e->stream));
        HIPCHECK(hipMemcpyAsync(e->nmoves_dev, n_moves, sizeof(int32_t) * n_slots, hipMemcpyHostToDevice, e->stream));
    }
    azx_launch_reset(d, slots ? e->slots_dev : nullptr, n_slots, moves ? e->moves_dev : nullptr,
                     moves ? e->nmoves_dev : nullptr, stride, 1, e->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(e->stream));
    e->ext_active = false;
    return AZX_OK;
}

// ---- timing of the tree kernels (roofline: algorithmic bytes / measured launch time) --------
static void time_begin(azx_engine *e, char tag = 0) {
    if (e->ev_used + 2 > e->ev_pool.size()) {
        if (e->ev_pool.size() >= 1 << 16) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        e->ev_pool.push_back(a);
        e->ev_pool.push_back(b);
    }
    if (e->ev_tag.size() < e->ev_pool.size() / 2) e->ev_tag.resize(e->ev_pool.size() / 2, 0);
    e->ev_tag[e->ev_used / 2] = tag;
    if (e->ev_weight.size() < e->ev_pool.size() / 2) e->ev_weight.resize(e->ev_pool.size() / 2, 1);
    e->ev_weight[e->ev_used / 2] = 1;
    (void)hipEventRecord(e->ev_pool[e->ev_used], e->stream);
}
static void time_end(azx_engine *e, int weight = 1) {
    if (e->ev_used + 2 > e->ev_pool.size()) return;     // pool full: time_begin recorded nothing either
    (void)hipEventRecord(e->ev_pool[e->ev_used + 1], e->stream);
    e->ev_weight[e->ev_used / 2] = weight;
    e->ev_used += 2;
}
static void time_collect(azx_engine *e, azx_play_stats *st) {
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e->ev_pool[i], e->ev_pool[i + 1]) == hipSuccess) {
            if (e->ev_tag[i / 2]) { st->net_seconds += ms * 1e-3; st->net_launches += 1; }
            else {
                st->mcts_seconds += ms * 1e-3;
                st->mcts_launches += e->ev_weight[i / 2];
                st->mcts_kernel_launches += 1;
            }
        }
    }
    e->ev_used = 0;
}

static int upload_noise(azx_engine *e, const double *noise, int n_select, int noise_stride,
                        double noise_scale) {
    DevEngine &d = e->d;
    d.noise = nullptr;
    d.device_noise = 0;
    d.noise_scale = noise_scale;
    d.n_select = n_select;
    d.noise_stride = noise_stride;
    if (noise_scale == 0.0) return AZX_OK;
    if (!noise) { d.device_noise = 1; return AZX_OK; }
    if (n_select < e->selects_per_search)
        return fail(AZX_EINVAL, "noise has %d rows, a search consumes %d", n_select, e->selects_per_search);
    const size_t need = (size_t)d.G * n_select * noise_stride;
    if (need > e->noise_cap) {
        if (e->noise_dev) {      // searches are blocking: no kernel still reads the old rows
            (void)hipFree(e->noise_dev);
            e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void *)e->noise_dev), e->allocs.end());
            e->noise_dev = nullptr;
            e->noise_cap = 0;
        }
        TRY(dev_alloc(e, &e->noise_dev, need, false));
        e->noise_cap = need;
    }
    HIPCHECK(hipMemcpyAsync(e->noise_dev, noise, sizeof(double) * need, hipMemcpyHostToDevice, e->stream));
    d.noise = e->noise_dev;
    return AZX_OK;
}

// one whole search on the stream (no host sync) for the device-side evaluators
static int enqueue_search(azx_engine *e, bool timed) {
    DevEngine &d = e->d;
    if (d.evaluator == AZX_EVAL_UNIFORM || d.evaluator == AZX_EVAL_UNIFORM_HASH) {
        if (timed) time_begin(e);
        azx_launch_mcts(d, MODE_BEGIN | MODE_INLINE, e->num_batches, e->stream, e->force_generic);
        if (timed) time_end(e);
        return AZX_OK;
    }
    if (d.evaluator == AZX_EVAL_RESNET) {
        if (!azx_net_ready(e->net)) return fail(AZX_ESTATE, "azx_set_weights has not been called");
        HIPCHECK(hipMemsetAsync(d.n_eval, 0, sizeof(int32_t), e->stream));
        if (timed) time_begin(e);
        azx_launch_mcts(d, MODE_BEGIN, e->num_batches, e->stream, e->force_generic);
        if (timed) time_end(e);
        if (timed) time_begin(e, 1);
        azx_net_eval(e->net, d, e->stream);
        if (timed) time_end(e);
        for (int b = 0; b < e->num_batches; ++b) {
            HIPCHECK(hipMemsetAsync(d.n_eval, 0, sizeof(int32_t), e->stream));
            if (timed) time_begin(e);
            azx_launch_mcts(d, MODE_APPLY | MODE_SELECT, e->num_batches, e->stream, e->force_generic);
            if (timed) time_end(e);
            if (timed) time_begin(e, 1);
            azx_net_eval(e->net, d, e->stream);
            if (timed) time_end(e);
        }
        if (timed) time_begin(e);
        azx_launch_mcts(d, MODE_APPLY, e->num_batches, e->stream, e->force_generic);
        if (timed) time_end(e);
        return AZX_OK;
    }
    return fail(AZX_ESTATE, "evaluator %d has no device pipeline", d.evaluator);
}

extern "C" int azx_search(azx_engine *e, const double *noise, int n_select, int noise_stride,
                          double noise_scale) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (e->d.evaluator == AZX_EVAL_EXTERNAL)
        return fail(AZX_ESTATE, "AZX_EVAL_EXTERNAL: drive azx_search_begin/step instead");
    TRY(upload_noise(e, noise, n_select, noise_stride, noise_scale));
    TRY(enqueue_search(e, false));
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(e->stream));
    return check_net_range(e);
}

static int read_pending(azx_engine *e, int *n_pending) {
    int32_t n = 0;
    HIPCHECK(hipMemcpyAsync(&n, e->d.n_eval, sizeof n, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    *n_pending = n;
    return e->d.evaluator == AZX_EVAL_RESNET ? check_net_range(e) : AZX_OK;
}

extern "C" int azx_search_begin(azx_engine *e, const double *noise, int n_select, int noise_stride,
                                double noise_scale, int *n_pending) {
    if (!e || !n_pending) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->d.evaluator != AZX_EVAL_EXTERNAL && e->d.evaluator != AZX_EVAL_RESNET)
        return fail(AZX_ESTATE, "phase API needs AZX_EVAL_EXTERNAL (or RESNET)");
    TRY(upload_noise(e, noise, n_select, noise_stride, noise_scale));
    HIPCHECK(hipMemsetAsync(e->d.n_eval, 0, sizeof(int32_t), e->stream));
    if (e->d.evaluator == AZX_EVAL_RESNET && !azx_net_ready(e->net))
        return fail(AZX_ESTATE, "azx_set_weights has not been called");
    azx_launch_mcts(e->d, MODE_BEGIN, e->num_batches, e->stream, e->force_generic);
    if (e->d.evaluator == AZX_EVAL_RESNET) azx_net_eval(e->net, e->d, e->stream);
    HIPCHECK(hipGetLastError());
    e->ext_active = true;
    e->ext_batches_done = 0;
    return read_pending(e, n_pending);
}

extern "C" int azx_search_step(azx_engine *e, int *n_pending, int *done) {
    if (!e || !n_pending || !done) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (!e->ext_active) return fail(AZX_ESTATE, "azx_search_step without azx_search_begin");
    HIPCHECK(hipMemsetAsync(e->d.n_eval, 0, sizeof(int32_t), e->stream));
    if (e->ext_batches_done < e->num_batches) {
        azx_launch_mcts(e->d, MODE_APPLY | MODE_SELECT, e->num_batches, e->stream, e->force_generic);
        if (e->d.evaluator == AZX_EVAL_RESNET) azx_net_eval(e->net, e->d, e->stream);
        e->ext_batches_done += 1;
        *done = 0;
    } else {
        azx_launch_mcts(e->d, MODE_APPLY, e->num_batches, e->stream, e->force_generic);
        e->ext_active = false;
        *done = 1;
    }
    HIPCHECK(hipGetLastError());
    return read_pending(e, n_pending);
}

static int flip_cell(int cell, int N) {   // hex.py:107-111 (r,c) -> (N-1-c, N-1-r)
    const int r = cell / N, c = cell % N;
    return (N - 1 - c) * N + (N - 1 - r);
}

extern "C" int azx_get_leaves(azx_engine *e, int cap, int32_t *boards, int32_t *legal_moves,
                              int32_t *slot, int32_t *k, int *n_out) {
    if (!e || !n_out) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    int n = 0;
    TRY(read_pending(e, &n));
    if (n > cap) return fail(AZX_EINVAL, "%d pending leaves, caller capacity %d", n, cap);
    *n_out = n;
    e->ext_order.clear();
    e->ext_cells.clear();
    if (n == 0) return AZX_OK;
    std::vector<uint8_t> hb((size_t)n * AZX_CELL_STRIDE);
    std::vector<int32_t> hsrc(n), hflip(n);
    HIPCHECK(hipMemcpyAsync(hb.data(), d.ev_board, hb.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipMemcpyAsync(hsrc.data(), d.ev_src, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipMemcpyAsync(hflip.data(), d.ev_flip, sizeof(int32_t) * n, hipMemcpyDeviceToHost, e->stream));
    const size_t E = (size_t)d.G * d.bs;
    std::vector<uint64_t> hmask(E * 4);
    HIPCHECK(hipMemcpyAsync(hmask.data(), d.leaf_mask, sizeof(uint64_t) * hmask.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hsrc[a] < hsrc[b]; });
    e->ext_order = order;
    e->ext_cells.resize(n);
    for (int j = 0; j < n; ++j) {
        const int ev = order[j], src = hsrc[ev];
        std::vector<int> &cells = e->ext_cells[j];
        for (int c = 0; c < d.ncells; ++c)
            if ((hmask[(size_t)src * 4 + (c >> 6)] >> (c & 63)) & 1ull) cells.push_back(c);
        if (slot) slot[j] = src / d.bs;
        if (k) k[j] = (int)cells.size();
        if (boards)
            for (int c = 0; c < d.ncells; ++c)
                boards[(size_t)j * d.ncells + c] = hb[(size_t)ev * AZX_CELL_STRIDE + c];
        if (legal_moves) {
            for (int c = 0; c < d.ncells; ++c) legal_moves[(size_t)j * d.ncells + c] = 0;
            for (size_t i = 0; i < cells.size(); ++i)
                legal_moves[(size_t)j * d.ncells + i] =
                    (hflip[ev] ? flip_cell(cells[i], d.N) : cells[i]) + 1;
        }
    }
    return AZX_OK;
}

extern "C" int azx_put_evals(azx_engine *e, int n, const float *value, const float *prior) {
    if (!e || (n && (!value || !prior))) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    if (n != (int)e->ext_order.size())
        return fail(AZX_ESTATE, "azx_put_evals(%d) does not match %zu pending leaves", n, e->ext_order.size());
    if (n == 0) return AZX_OK;
    std::vector<float> hv(n), hp((size_t)n * AZX_CELL_STRIDE, 0.0f);
    for (int j = 0; j < n; ++j) {
        const int ev = e->ext_order[j];
        hv[ev] = value[j];
        const std::vector<int> &cells = e->ext_cells[j];
        for (size_t i = 0; i < cells.size(); ++i)
            hp[(size_t)ev * AZX_CELL_STRIDE + cells[i]] = prior[(size_t)j * d.ncells + i];
    }
    HIPCHECK(hipMemcpyAsync(d.ev_value, hv.data(), sizeof(float) * n, hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(d.ev_prior, hp.data(), sizeof(float) * hp.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_get_evals(azx_engine *e, int cap, float *value, float *prior, int *n_out) {
    if (!e || !n_out) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    const int n = (int)e->ext_order.size();
    if (n > cap) return fail(AZX_EINVAL, "%d evaluations, caller capacity %d", n, cap);
    *n_out = n;
    if (n == 0) return AZX_OK;
    std::vector<float> hv(n), hp((size_t)n * AZX_CELL_STRIDE);
    HIPCHECK(hipMemcpyAsync(hv.data(), d.ev_value, sizeof(float) * n, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipMemcpyAsync(hp.data(), d.ev_prior, sizeof(float) * hp.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    for (int j = 0; j < n; ++j) {
        const int ev = e->ext_order[j];
        if (value) value[j] = hv[ev];
        if (prior) {
            const std::vector<int> &cells = e->ext_cells[j];
            for (int c = 0; c < d.ncells; ++c) prior[(size_t)j * d.ncells + c] = 0.0f;
            for (size_t i = 0; i < cells.size(); ++i)
                prior[(size_t)j * d.ncells + i] = hp[(size_t)ev * AZX_CELL_STRIDE + cells[i]];
        }
    }
    return AZX_OK;
}

extern "C" int azx_get_root(azx_engine *e, int32_t *k, int32_t *legal_moves, float *child_visits,
                            float *child_value, float *child_prior, float *root_visits,
                            float *root_value, int32_t *num_nodes, float *search_value) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    const size_t G = d.G, GC = G * d.ncells;
    HIPCHECK(hipMemsetAsync(e->g_legal, 0, sizeof(int32_t) * GC, e->stream));
    HIPCHECK(hipMemsetAsync(e->g_cv, 0, sizeof(float) * GC, e->stream));
    HIPCHECK(hipMemsetAsync(e->g_cw, 0, sizeof(float) * GC, e->stream));
    HIPCHECK(hipMemsetAsync(e->g_cp, 0, sizeof(float) * GC, e->stream));
    azx_launch_gather_root(d, e->g_k, e->g_legal, e->g_cv, e->g_cw, e->g_cp, e->g_rv, e->g_rw,
                           e->g_nn, e->g_sv, e->stream);
    HIPCHECK(hipGetLastError());
#define D2H(dst, src, bytes) if (dst) HIPCHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream))
    D2H(k, e->g_k, sizeof(int32_t) * G);
    D2H(legal_moves, e->g_legal, sizeof(int32_t) * GC);
    D2H(child_visits, e->g_cv, sizeof(float) * GC);
    D2H(child_value, e->g_cw, sizeof(float) * GC);
    D2H(child_prior, e->g_cp, sizeof(float) * GC);
    D2H(root_visits, e->g_rv, sizeof(float) * G);
    D2H(root_value, e->g_rw, sizeof(float) * G);
    D2H(num_nodes, e->g_nn, sizeof(int32_t) * G);
    D2H(search_value, e->g_sv, sizeof(float) * G);
#undef D2H
    HIPCHECK(hipStreamSynchronize(e->stream));
    if (search_value) {   // mcts.py:291
        const float denom = (float)(e->num_batches * d.bs);
        for (size_t g = 0; g < G; ++g) search_value[g] = search_value[g] / denom;
    }
    return AZX_OK;
}

extern "C" int azx_get_status(azx_engine *e, int32_t *status) {
    if (!e || !status) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    std::vector<TreeHdr> th(e->d.G);
    HIPCHECK(hipMemcpyAsync(th.data(), e->d.thdr, sizeof(TreeHdr) * th.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    for (int g = 0; g < e->d.G; ++g) status[g] = th[g].status;
    return AZX_OK;
}

extern "C" int azx_get_tree_nodes(azx_engine *e, int32_t *nodes) {
    if (!e || !nodes) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    std::vector<TreeHdr> th(e->d.G);
    HIPCHECK(hipMemcpyAsync(th.data(), e->d.thdr, sizeof(TreeHdr) * th.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    for (int g = 0; g < e->d.G; ++g) nodes[g] = th[g].num_nodes + th[g].dropped;
    return AZX_OK;
}

extern "C" int azx_get_games(azx_engine *e, int32_t *board, int32_t *color, int32_t *result,
                             int32_t *ply) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    const size_t G = d.G;
    std::vector<uint32_t> hc(G * d.slots * 64);
    std::vector<GameHdr> hh(G);
    HIPCHECK(hipMemcpyAsync(hc.data(), d.cells, sizeof(uint32_t) * hc.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipMemcpyAsync(hh.data(), d.ghdr, sizeof(GameHdr) * G, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    for (size_t g = 0; g < G; ++g) {
        if (board)
            for (int c = 0; c < d.ncells; ++c)
                board[g * d.ncells + c] = (int32_t)(hc[g * d.slots * 64 + c] & 3u);
        if (color) color[g] = hh[g].color - 1;                                    // hex.py:56
        if (result) result[g] = hh[g].winner ? (hh[g].winner == 2 ? 1 : 3) : 0;   // hex.py:161-170
        if (ply) ply[g] = hh[g].ply;
    }
    return AZX_OK;
}

extern "C" int azx_advance(azx_engine *e, const int32_t *move_ids) {
    if (!e || !move_ids) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    HIPCHECK(hipMemcpyAsync(e->moveids_dev, move_ids, sizeof(int32_t) * d.G, hipMemcpyHostToDevice, e->stream));
    azx_launch_advance(d, e->moveids_dev, 0, e->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(e->stream));
    e->ext_active = false;
    return AZX_OK;
}

extern "C" int azx_tree_dump(azx_engine *e, int slot, int cap, int32_t *parent, int32_t *first_child,
                             int32_t *num_children, float *num_visits, float *total_value,
                             float *prior_prob, int32_t *num_nodes, int32_t *root_id) {
    if (!e || !num_nodes || !root_id) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    if (slot < 0 || slot >= d.G) return fail(AZX_EINVAL, "slot %d out of range", slot);
    TreeHdr th;
    HIPCHECK(hipMemcpyAsync(&th, d.thdr + slot, sizeof th, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    *num_nodes = th.num_nodes;
    *root_id = th.root_id;
    if (th.num_nodes > cap) return fail(AZX_EINVAL, "tree has %d nodes, caller capacity %d", th.num_nodes, cap);
    std::vector<Node> nodes(th.num_nodes);
    HIPCHECK(hipMemcpyAsync(nodes.data(), d.arena[th.arena] + (size_t)slot * d.cap,
                            sizeof(Node) * nodes.size(), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    // rebuild the reference's six arrays (search_tree.py:48-55): parent and num_children are
    // implied by the layout (children of a k-move node are k consecutive ids, each with k-1)
    const int n = th.num_nodes;
    std::vector<int32_t> par(n, -1), kk(n, -1);
    kk[0] = th.k0;
    for (int v = 0; v < n; ++v) {
        const Node &nd = nodes[v];
        int fc = -1, nc = -1;
        if (nd.link >= 0) {
            fc = nd.link;
            nc = kk[v];
            for (int j = 0; j < nc && fc + j < n; ++j) { par[fc + j] = v; kk[fc + j] = nc - 1; }
        } else if (nd.link <= -2) {
            fc = -2 - nd.link;
            nc = 0;
        }
        if (first_child) first_child[v] = fc;
        if (num_children) num_children[v] = nc;
        if (num_visits) num_visits[v] = nd.nv;
        if (total_value) total_value[v] = nd.tv;
        if (prior_prob) prior_prob[v] = nd.pp;
    }
    if (parent) for (int v = 0; v < n; ++v) parent[v] = par[v];
    return AZX_OK;
}

extern "C" int azx_forward(azx_engine *e, int B, int K, const int32_t *boards,
                           const int32_t *legal_moves, float *value, float *moves_logprob) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (!e->net) return fail(AZX_ESTATE, "engine was not created with AZX_EVAL_RESNET");
    int rc = azx_net_forward_host(e->net, B, K, boards, legal_moves, value, moves_logprob, e->stream);
    if (rc) g_err = azx_net_error();
    return rc ? rc : check_net_range(e);
}

extern "C" int azx_hex_replay(int device, int board_size, int n_games, const int32_t *moves,
                              const int32_t *length, int stride, int32_t *result_out,
                              int32_t *nlegal_out, uint64_t *empties_out, int32_t *final_board) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AZX_ENODEV, "no HIP device visible: the engine has no CPU fallback");
    if (board_size < 2 || board_size > AZX_MAX_BOARD) return fail(AZX_EINVAL, "bad board_size");
    if (n_games < 1 || stride < 1 || !moves || !length) return fail(AZX_EINVAL, "bad argument");
    DevGuard guard(device);
    if (azx_init_geometry(device)) return fail(AZX_EHIP, "uploading the board geometry tables failed");
    const int ncells = board_size * board_size;
    for (int g = 0; g < n_games; ++g) {
        if (length[g] < 0 || length[g] > stride) return fail(AZX_EINVAL, "length[%d] out of range", g);
        std::vector<char> used(ncells, 0);
        for (int p = 0; p < length[g]; ++p) {
            const int mv = moves[(size_t)g * stride + p];
            if (mv < 1 || mv > ncells || used[mv - 1]) return fail(AZX_EINVAL, "illegal move %d in game %d", mv, g);
            used[mv - 1] = 1;
        }
    }
    const size_t gs = (size_t)n_games * stride;
    int32_t *dm = nullptr, *dl = nullptr, *dr = nullptr, *dn = nullptr, *df = nullptr;
    uint64_t *de = nullptr;
    HIPCHECK(hipMalloc(&dm, sizeof(int32_t) * gs));
    HIPCHECK(hipMalloc(&dl, sizeof(int32_t) * n_games));
    HIPCHECK(hipMalloc(&dr, sizeof(int32_t) * gs));
    HIPCHECK(hipMalloc(&dn, sizeof(int32_t) * gs));
    HIPCHECK(hipMalloc(&de, sizeof(uint64_t) * gs * 4));
    HIPCHECK(hipMalloc(&df, sizeof(int32_t) * (size_t)n_games * ncells));
    HIPCHECK(hipMemset(dr, 0, sizeof(int32_t) * gs));
    HIPCHECK(hipMemset(dn, 0, sizeof(int32_t) * gs));
    HIPCHECK(hipMemset(de, 0, sizeof(uint64_t) * gs * 4));
    HIPCHECK(hipMemcpy(dm, moves, sizeof(int32_t) * gs, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dl, length, sizeof(int32_t) * n_games, hipMemcpyHostToDevice));
    azx_launch_hex_replay(board_size, n_games, dm, dl, stride, dr, dn, de, df, nullptr);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipDeviceSynchronize());
    if (result_out) HIPCHECK(hipMemcpy(result_out, dr, sizeof(int32_t) * gs, hipMemcpyDeviceToHost));
    if (nlegal_out) HIPCHECK(hipMemcpy(nlegal_out, dn, sizeof(int32_t) * gs, hipMemcpyDeviceToHost));
    if (empties_out) HIPCHECK(hipMemcpy(empties_out, de, sizeof(uint64_t) * gs * 4, hipMemcpyDeviceToHost));
    if (final_board) HIPCHECK(hipMemcpy(final_board, df, sizeof(int32_t) * (size_t)n_games * ncells, hipMemcpyDeviceToHost));
    (void)hipFree(dm); (void)hipFree(dl); (void)hipFree(dr); (void)hipFree(dn); (void)hipFree(de); (void)hipFree(df);
    return AZX_OK;
}

// ---- which slots take part in the next searches (tournaments: only the games whose turn it is) --
extern "C" int azx_set_active(azx_engine *e, const int32_t *active) {
    if (!e || !active) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    std::vector<int32_t> a((size_t)d.G);
    for (int g = 0; g < d.G; ++g) a[g] = active[g] ? 1 : 0;
    HIPCHECK(hipMemcpy2DAsync(&d.ghdr[0].active, sizeof(GameHdr), a.data(), sizeof(int32_t), sizeof(int32_t),
                              (size_t)d.G, hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

// ---- throughput mode ------------------------------------------------------------------------
static int play_setup(azx_engine *e, int64_t q_rows, int ring) {
    DevEngine &d = e->d;
    e->q_rows_valid = 0;
    if (!e->play_ready) {
        const size_t rows = (size_t)d.G * d.ncells;
        TRY(dev_alloc(e, &d.row_board, rows * AZX_CELL_STRIDE));
        TRY(dev_alloc(e, &d.row_prob, rows * AZX_CELL_STRIDE));
        TRY(dev_alloc(e, &d.row_k, rows));
        TRY(dev_alloc(e, &d.row_meta, rows * AZX_ROW_METRICS));
        e->play_ready = true;
    }
    if (q_rows > e->q_alloc) {
        (void)hipStreamSynchronize(e->stream);
        for (void *p : e->q_allocs) (void)hipFree(p);
        e->q_allocs.clear();
        auto qa = [&](void **p, size_t bytes) -> int {
            hipError_t err = hipMalloc(p, bytes);
            if (err != hipSuccess) return fail(AZX_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
            e->q_allocs.push_back(*p);
            return AZX_OK;
        };
        TRY(qa((void **)&d.q_board, (size_t)q_rows * AZX_CELL_STRIDE));
        TRY(qa((void **)&d.q_prob, (size_t)q_rows * AZX_CELL_STRIDE * sizeof(float)));
        TRY(qa((void **)&d.q_color, (size_t)q_rows * sizeof(int32_t)));
        TRY(qa((void **)&d.q_k, (size_t)q_rows * sizeof(int32_t)));
        TRY(qa((void **)&d.q_reward, (size_t)q_rows * sizeof(float)));
        TRY(qa((void **)&d.q_uid, (size_t)q_rows * sizeof(int64_t)));
        TRY(qa((void **)&d.q_meta, (size_t)q_rows * AZX_ROW_METRICS * sizeof(float)));
        e->q_alloc = q_rows;
    }
    d.q_cap = e->q_alloc;
    d.q_ring = ring;
    HIPCHECK(hipMemsetAsync(d.q_count, 0, sizeof(unsigned long long), e->stream));
    return AZX_OK;
}

struct CounterSnap {
    unsigned long long c[CTR_COUNT];
    double s[8];
};

static int snap_counters(azx_engine *e, CounterSnap *s) {
    const size_t G = e->d.G;
    std::vector<unsigned long long> hc(G * CTR_COUNT);
    std::vector<double> hs(G * 8);
    HIPCHECK(hipMemcpyAsync(hc.data(), e->d.counters, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipMemcpyAsync(hs.data(), e->d.stat_sums, hs.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    memset(s, 0, sizeof *s);
    for (size_t g = 0; g < G; ++g) {   // per-game accumulators (no device atomics): sum here
        for (int j = 0; j < CTR_COUNT; ++j) s->c[j] += hc[g * CTR_COUNT + j];
        for (int j = 0; j < 8; ++j) s->s[j] += hs[g * 8 + j];
    }
    return AZX_OK;
}

static void fill_stats(const CounterSnap &a, const CounterSnap &b, azx_play_stats *st) {
    st->games = (int64_t)(b.c[CTR_GAMES] - a.c[CTR_GAMES]);
    st->game_errors = (int64_t)(b.c[CTR_ERRORS] - a.c[CTR_ERRORS]);
    st->plies = (int64_t)(b.c[CTR_PLIES] - a.c[CTR_PLIES]);
    st->selects = (int64_t)(b.c[CTR_SELECTS] - a.c[CTR_SELECTS]);
    st->evals = (int64_t)(b.c[CTR_EVALS] - a.c[CTR_EVALS]);
    st->sum_depth = (int64_t)(b.c[CTR_SUM_DEPTH] - a.c[CTR_SUM_DEPTH]);
    st->sum_k_interior = (int64_t)(b.c[CTR_SUM_K_INT] - a.c[CTR_SUM_K_INT]);
    st->sum_k_leaf = (int64_t)(b.c[CTR_SUM_K_LEAF] - a.c[CTR_SUM_K_LEAF]);
    st->sum_search_value = b.s[0] - a.s[0];
    st->sum_root_width = b.s[1] - a.s[1];
    st->sum_action_logprob = b.s[2] - a.s[2];
    st->sum_reward_last = b.s[3] - a.s[3];
    st->sum_game_length = b.s[4] - a.s[4];
}

static int enqueue_ply(azx_engine *e) {
    TRY(enqueue_search(e, true));
    azx_launch_choose(e->d, e->stream);
    azx_launch_advance(e->d, nullptr, 1, e->stream);
    return AZX_OK;
}

extern "C" int azx_play_steps(azx_engine *e, int64_t plies, azx_play_stats *stats) {
    if (!e || !stats) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->d.evaluator == AZX_EVAL_EXTERNAL) return fail(AZX_ESTATE, "play mode needs a device evaluator");
    memset(stats, 0, sizeof *stats);
    TRY(play_setup(e, std::max<int64_t>(e->q_alloc, 1 << 16), 1));
    TRY(upload_noise(e, nullptr, 0, 0, e->cfg.noise_scale));
    CounterSnap a, b;
    TRY(snap_counters(e, &a));
    hipEvent_t t0, t1;
    HIPCHECK(hipEventCreate(&t0));
    HIPCHECK(hipEventCreate(&t1));
    HIPCHECK(hipEventRecord(t0, e->stream));
    azx_launch_advance(e->d, nullptr, 2, e->stream);     // slots parked by an earlier azx_play* call rejoin (ring queue: always room)
    {
        // the uniform-evaluator path plays the moves in persistent launches (k_play), at most
        // PLAY_CHUNK moves each; its time is booked per move like the per-move launches'
        const int PLAY_CHUNK = 256;
        int64_t p = 0;
        while (p < plies) {
            const int n = (int)std::min<int64_t>(PLAY_CHUNK, plies - p);
            time_begin(e);
            const bool ok = azx_launch_play(e->d, e->num_batches, n, e->stream, !e->force_generic && !e->no_persistent);
            if (!ok) break;                 // (the begin event is simply overwritten by the next one)
            time_end(e, n);
            p += n;
        }
        for (; p < plies; ++p) TRY(enqueue_ply(e));
    }
    HIPCHECK(hipEventRecord(t1, e->stream));
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    TRY(snap_counters(e, &b));
    fill_stats(a, b, stats);
    stats->positions = (int64_t)(b.c[CTR_ROWS] - a.c[CTR_ROWS]);
    stats->seconds = ms * 1e-3;
    time_collect(e, stats);
    return check_net_range(e);
}

// play whole games into the harvest queue until it holds >= min_positions rows
static int play_until(azx_engine *e, int64_t min_positions, int64_t max_plies, azx_play_stats *stats,
                      unsigned long long *rows_out) {
    DevEngine &d = e->d;
    memset(stats, 0, sizeof *stats);
    const int64_t worst = min_positions + (int64_t)d.G * d.ncells;
    TRY(play_setup(e, worst, 0));
    if (e->dbg_qcap > 0)                                // tests (azx_debug_set_queue_cap): a queue too small, so that slots get parked
        d.q_cap = std::max<int64_t>(1, std::min<int64_t>(d.q_cap, e->dbg_qcap));
    TRY(upload_noise(e, nullptr, 0, 0, e->cfg.noise_scale));
    CounterSnap a, b;
    TRY(snap_counters(e, &a));
    hipEvent_t t0, t1;
    HIPCHECK(hipEventCreate(&t0));
    HIPCHECK(hipEventCreate(&t1));
    HIPCHECK(hipEventRecord(t0, e->stream));
    // games that finished while the queue was full (parked slots) hand their rows over first
    azx_launch_advance(d, nullptr, 2, e->stream);
    unsigned long long rows = 0;
    for (int64_t p = 0; (max_plies <= 0 || p < max_plies) && (int64_t)rows < min_positions;) {
        // with the uniform evaluator a few moves per persistent launch (k_play) between looks at the
        // queue; Player.read may return more than it was asked for anyway (whole games only)
        // (at most 2N - 1 moves, the shortest possible game: a slot then finishes at most one game
        // per launch and the queue bound min_positions + n_games * cells still holds)
        const int64_t most = std::min<int64_t>(8, 2 * d.N - 1);
        const int chunk = (int)std::min<int64_t>(most, max_plies > 0 ? max_plies - p : most);
        time_begin(e);
        if (azx_launch_play(d, e->num_batches, chunk, e->stream, !e->force_generic && !e->no_persistent)) {
            time_end(e, chunk);
            p += chunk;
        } else {
            TRY(enqueue_ply(e));
            p += 1;
        }
        HIPCHECK(hipMemcpyAsync(&rows, d.q_count, sizeof rows, hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
        // a bounded queue (azx_debug_set_queue_cap) below min_positions parks every slot sooner or later: with
        // no slot left to play, return what the queue holds instead of spinning
        if (e->dbg_qcap > 0 && (p & 7) == 0) {
            std::vector<GameHdr> hh((size_t)d.G);
            HIPCHECK(hipMemcpyAsync(hh.data(), d.ghdr, sizeof(GameHdr) * hh.size(), hipMemcpyDeviceToHost, e->stream));
            HIPCHECK(hipStreamSynchronize(e->stream));
            bool any = false;
            for (const GameHdr &h : hh) any = any || h.active;
            if (!any) break;
        }
    }
    HIPCHECK(hipEventRecord(t1, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, t0, t1));
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    TRY(snap_counters(e, &b));
    fill_stats(a, b, stats);
    stats->positions = (int64_t)rows;
    stats->seconds = ms * 1e-3;
    time_collect(e, stats);
    *rows_out = rows;
    return check_net_range(e);
}

extern "C" int azx_play(azx_engine *e, int64_t min_positions, int64_t max_plies, int64_t cap,
                        int32_t *board, int32_t *color, int32_t *nlegal, float *moves_prob,
                        float *reward, int64_t *game_uid, azx_play_stats *stats) {
    if (!e || !stats) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->d.evaluator == AZX_EVAL_EXTERNAL) return fail(AZX_ESTATE, "play mode needs a device evaluator");
    DevEngine &d = e->d;
    const int64_t worst = min_positions + (int64_t)d.G * d.ncells;
    if (cap < worst)
        return fail(AZX_EINVAL, "cap %lld < min_positions + n_games*cells = %lld (whole games only)",
                    (long long)cap, (long long)worst);
    unsigned long long rows = 0;
    TRY(play_until(e, min_positions, max_plies, stats, &rows));
    e->q_rows_valid = (int64_t)rows;
    if (rows == 0) return AZX_OK;
    const size_t n = (size_t)rows;
    // widen / densify on the device (k_rows_export), then one copy per output array
    if (board || moves_prob) {
        const size_t need = n * d.ncells;
        if (need > e->export_cap) {
            (void)hipStreamSynchronize(e->stream);
            if (e->export_board) (void)hipFree(e->export_board);
            if (e->export_prob) (void)hipFree(e->export_prob);
            e->export_board = nullptr; e->export_prob = nullptr; e->export_cap = 0;
            if (hipMalloc((void **)&e->export_board, need * sizeof(int32_t)) != hipSuccess ||
                hipMalloc((void **)&e->export_prob, need * sizeof(float)) != hipSuccess)
                return fail(AZX_ENOMEM, "hipMalloc of the export staging (%zu rows) failed", n);
            e->export_cap = need;
        }
        azx_launch_rows_export(d.q_board, d.q_prob, (long long)n, d.ncells, e->export_board, e->export_prob, e->stream);
        HIPCHECK(hipGetLastError());
        if (board) HIPCHECK(hipMemcpyAsync(board, e->export_board, need * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
        if (moves_prob) HIPCHECK(hipMemcpyAsync(moves_prob, e->export_prob, need * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    }
    if (color) HIPCHECK(hipMemcpyAsync(color, d.q_color, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (nlegal) HIPCHECK(hipMemcpyAsync(nlegal, d.q_k, n * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (reward) HIPCHECK(hipMemcpyAsync(reward, d.q_reward, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (game_uid) HIPCHECK(hipMemcpyAsync(game_uid, d.q_uid, n * sizeof(int64_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_play_device(azx_engine *e, int64_t min_positions, int64_t max_plies, int64_t *rows_out,
                               azx_play_stats *stats) {
    if (!e || !stats || !rows_out) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->d.evaluator == AZX_EVAL_EXTERNAL) return fail(AZX_ESTATE, "play mode needs a device evaluator");
    unsigned long long rows = 0;
    TRY(play_until(e, min_positions, max_plies, stats, &rows));
    e->q_rows_valid = (int64_t)rows;
    *rows_out = (int64_t)rows;
    return AZX_OK;
}

extern "C" int azx_play_row_metrics(azx_engine *e, int64_t cap, float *metrics, int64_t *n_out) {
    if (!e || !metrics || !n_out) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    const int64_t n = e->q_rows_valid;
    if (n > cap) return fail(AZX_EINVAL, "%lld rows queued, caller capacity %lld", (long long)n, (long long)cap);
    *n_out = n;
    if (n == 0) return AZX_OK;
    HIPCHECK(hipMemcpyAsync(metrics, e->d.q_meta, (size_t)n * AZX_ROW_METRICS * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_rows_pack(azx_engine *e, int64_t first, int64_t n, void *records_dev) {
    if (!e || (n > 0 && !records_dev)) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (first < 0 || n < 0 || first + n > e->q_rows_valid)
        return fail(AZX_EINVAL, "rows [%lld, %lld) outside the %lld rows queued", (long long)first,
                    (long long)(first + n), (long long)e->q_rows_valid);
    if (n == 0) return AZX_OK;
    DevEngine &d = e->d;
    const ReplayRows src = {d.q_board, d.q_prob, d.q_color, d.q_k, d.q_reward};
    azx_launch_rows_pack(src, (const long long *)d.q_uid, first, n, d.ncells, (uint8_t *)records_dev, e->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_replay_put_records(azx_engine *e, int64_t n, const void *records_dev) {
    if (!e || (n > 0 && !records_dev)) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (n < 0) return fail(AZX_EINVAL, "n must be >= 0");
    if (n == 0) return AZX_OK;
    azx_launch_records_put((const uint8_t *)records_dev, e->ring, n, e->ring_cap, e->ring_write, e->d.ncells, e->stream);
    HIPCHECK(hipGetLastError());
    e->ring_write = (e->ring_write + n) % e->ring_cap;
    e->ring_size = std::min<int64_t>(e->ring_cap, e->ring_size + n);
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_replay_put_records_async(azx_engine *e, int64_t n, const void *records_dev, void *hip_stream) {
    if (!e || (n > 0 && !records_dev)) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (n < 0) return fail(AZX_EINVAL, "n must be >= 0");
    if (n == 0) return AZX_OK;
    // only ring state and the caller's stream: nothing the play thread (azx_play_device on e->stream) touches
    azx_launch_records_put((const uint8_t *)records_dev, e->ring, n, e->ring_cap, e->ring_write, e->d.ncells, (hipStream_t)hip_stream);
    HIPCHECK(hipGetLastError());
    e->ring_write = (e->ring_write + n) % e->ring_cap;
    e->ring_size = std::min<int64_t>(e->ring_cap, e->ring_size + n);
    return AZX_OK;
}

// ---- device-resident replay ring: ReplayBuffer.put FIFO (replay_buffer.py:134-149) and the
// prep.batch_replays collate (prep.py:24-39) without leaving HBM ---------------------------------
extern "C" int azx_replay_create(azx_engine *e, int64_t capacity) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (capacity < 1) return fail(AZX_EINVAL, "capacity must be >= 1");
    (void)hipStreamSynchronize(e->stream);
    for (void *p : e->ring_allocs) (void)hipFree(p);
    e->ring_allocs.clear();
    e->ring_cap = e->ring_size = e->ring_write = 0;
    e->ring_idx = nullptr;
    e->ring_idx_cap = 0;
    auto ra = [&](void **p, size_t bytes) -> int {
        hipError_t err = hipMalloc(p, bytes);
        if (err != hipSuccess) return fail(AZX_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
        e->ring_allocs.push_back(*p);
        return AZX_OK;
    };
    TRY(ra((void **)&e->ring.board, (size_t)capacity * AZX_CELL_STRIDE));
    TRY(ra((void **)&e->ring.prob, (size_t)capacity * AZX_CELL_STRIDE * sizeof(float)));
    TRY(ra((void **)&e->ring.color, (size_t)capacity * sizeof(int32_t)));
    TRY(ra((void **)&e->ring.k, (size_t)capacity * sizeof(int32_t)));
    TRY(ra((void **)&e->ring.reward, (size_t)capacity * sizeof(float)));
    TRY(ra((void **)&e->ring_maxk, sizeof(int32_t)));
    e->ring_cap = capacity;
    return AZX_OK;
}

extern "C" int azx_replay_state(azx_engine *e, int64_t *capacity, int64_t *size, int64_t *write_idx) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (capacity) *capacity = e->ring_cap;
    if (size) *size = e->ring_size;
    if (write_idx) *write_idx = e->ring_write;
    return AZX_OK;
}

extern "C" int azx_replay_set_state(azx_engine *e, int64_t size, int64_t write_idx) {
    if (!e) return fail(AZX_EINVAL, "null engine");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (size < 0 || size > e->ring_cap || write_idx < 0 || write_idx >= e->ring_cap)
        return fail(AZX_EINVAL, "size/write_idx outside the ring");
    e->ring_size = size;
    e->ring_write = write_idx;
    return AZX_OK;
}

// rows [0, n) of `src` enter the ring in order, oldest rows overwritten first
static int ring_put(azx_engine *e, const ReplayRows &src, int64_t n) {
    azx_launch_replay_put(src, e->ring, n, e->ring_cap, e->ring_write, e->stream);
    HIPCHECK(hipGetLastError());
    e->ring_write = (e->ring_write + n) % e->ring_cap;
    e->ring_size = std::min<int64_t>(e->ring_cap, e->ring_size + n);
    return AZX_OK;
}

extern "C" int azx_replay_put(azx_engine *e, int64_t n, const int32_t *board, const int32_t *color,
                              const int32_t *nlegal, const float *moves_prob, const float *reward) {
    if (!e || !board || !color || !nlegal || !moves_prob || !reward) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (n < 0) return fail(AZX_EINVAL, "n must be >= 0");
    if (n == 0) return AZX_OK;
    DevEngine &d = e->d;
    // stage through the harvest queue buffers (same row format)
    TRY(play_setup(e, n, 0));
    std::vector<uint8_t> hb((size_t)n * AZX_CELL_STRIDE, 0);
    std::vector<float> hp((size_t)n * AZX_CELL_STRIDE, 0.0f);
    for (int64_t r = 0; r < n; ++r) {
        if (nlegal[r] < 0 || nlegal[r] > d.ncells) return fail(AZX_EINVAL, "row %lld: nlegal out of range", (long long)r);
        for (int c = 0; c < d.ncells; ++c) {
            const int32_t v = board[r * d.ncells + c];
            if (v < 0 || v > 2) return fail(AZX_EINVAL, "row %lld: cell value %d", (long long)r, v);
            hb[r * AZX_CELL_STRIDE + c] = (uint8_t)v;
        }
        for (int c = 0; c < nlegal[r]; ++c) hp[r * AZX_CELL_STRIDE + c] = moves_prob[r * d.ncells + c];
    }
    HIPCHECK(hipMemcpyAsync(d.q_board, hb.data(), hb.size(), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(d.q_prob, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(d.q_color, color, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(d.q_k, nlegal, n * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemcpyAsync(d.q_reward, reward, n * sizeof(float), hipMemcpyHostToDevice, e->stream));
    const ReplayRows src = {d.q_board, d.q_prob, d.q_color, d.q_k, d.q_reward};
    TRY(ring_put(e, src, n));
    HIPCHECK(hipStreamSynchronize(e->stream));   // the host staging vectors die here
    return AZX_OK;
}

extern "C" int azx_replay_fill(azx_engine *e, int64_t min_positions, int64_t max_plies,
                               int64_t *rows_out, azx_play_stats *stats) {
    if (!e || !stats) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (e->d.evaluator == AZX_EVAL_EXTERNAL) return fail(AZX_ESTATE, "play mode needs a device evaluator");
    unsigned long long rows = 0;
    TRY(play_until(e, min_positions, max_plies, stats, &rows));
    e->q_rows_valid = (int64_t)rows;
    DevEngine &d = e->d;
    const ReplayRows src = {d.q_board, d.q_prob, d.q_color, d.q_k, d.q_reward};
    if (rows) TRY(ring_put(e, src, (int64_t)rows));
    HIPCHECK(hipStreamSynchronize(e->stream));
    if (rows_out) *rows_out = (int64_t)rows;
    return AZX_OK;
}

extern "C" int azx_replay_collate(azx_engine *e, int64_t batch, const int64_t *indices, int64_t *color_dev,
                                  int32_t *legal_moves_dev, int64_t *result_dev, int32_t *board_dev,
                                  float *moves_prob_dev, float *reward_dev, int32_t *max_k_out) {
    if (!e || !indices || !color_dev || !legal_moves_dev || !result_dev || !board_dev || !moves_prob_dev ||
        !reward_dev || !max_k_out)
        return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (batch < 1 || batch > (1 << 24)) return fail(AZX_EINVAL, "batch outside [1, 2^24]");
    for (int64_t b = 0; b < batch; ++b)
        if (indices[b] < 0 || indices[b] >= e->ring_size)
            return fail(AZX_EINVAL, "index %lld outside the %lld rows held", (long long)indices[b], (long long)e->ring_size);
    if (batch > e->ring_idx_cap) {
        void *p = nullptr;
        hipError_t err = hipMalloc(&p, (size_t)batch * sizeof(long long));
        if (err != hipSuccess) return fail(AZX_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(err));
        if (e->ring_idx) {                 // blocking calls: no kernel still reads the old index buffer
            (void)hipFree(e->ring_idx);
            e->ring_allocs.erase(std::remove(e->ring_allocs.begin(), e->ring_allocs.end(), (void *)e->ring_idx),
                                 e->ring_allocs.end());
        }
        e->ring_allocs.push_back(p);
        e->ring_idx = (long long *)p;
        e->ring_idx_cap = batch;
    }
    static_assert(sizeof(long long) == sizeof(int64_t), "index width");
    HIPCHECK(hipMemcpyAsync(e->ring_idx, indices, (size_t)batch * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
    HIPCHECK(hipMemsetAsync(e->ring_maxk, 0, sizeof(int32_t), e->stream));
    azx_launch_replay_collate(e->ring, e->ring_idx, (int)batch, e->d.ncells, (long long *)color_dev,
                              legal_moves_dev, (long long *)result_dev, board_dev, moves_prob_dev, reward_dev,
                              e->ring_maxk, e->ring_mover_view ? e->d.N : 0, e->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(max_k_out, e->ring_maxk, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

// The same collate, enqueued on the CALLER's stream and not synchronised (no max_k: the consumer takes full-width
// rows): for a trainer whose step runs on that stream (azx_train_step) -- the host queues collate + step and moves on.
extern "C" int azx_replay_set_mover_view(azx_engine *e, int on) {
    if (!e) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    e->ring_mover_view = on != 0;
    return AZX_OK;
}

// The sampled indices are staged through a small ring of pinned buffers; a slot is re-used only after the collate that
// read it has run (one event per slot).  The caller orders ring WRITES (azx_replay_fill / put, which are blocking calls
// on the engine's stream) after these reads by synchronising its stream before a refill.
extern "C" int azx_replay_collate_async(azx_engine *e, int64_t batch, const int64_t *indices, int64_t *color_dev,
                                        int32_t *legal_moves_dev, int64_t *result_dev, int32_t *board_dev,
                                        float *moves_prob_dev, float *reward_dev, void *hip_stream) {
    if (!e || !indices || !color_dev || !legal_moves_dev || !result_dev || !board_dev || !moves_prob_dev || !reward_dev)
        return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (e->ring_cap == 0) return fail(AZX_ESTATE, "no replay ring: call azx_replay_create first");
    if (batch < 1 || batch > (1 << 20)) return fail(AZX_EINVAL, "batch outside [1, 2^20]");
    for (int64_t b = 0; b < batch; ++b)
        if (indices[b] < 0 || indices[b] >= e->ring_size)
            return fail(AZX_EINVAL, "index %lld outside the %lld rows held", (long long)indices[b], (long long)e->ring_size);
    hipStream_t st = (hipStream_t)hip_stream;
    if (batch > e->cidx_cap) {
        for (int i = 0; i < 8; ++i) {
            if (e->cidx_ev[i]) HIPCHECK(hipEventSynchronize(e->cidx_ev[i]));
            if (e->cidx_host[i]) (void)hipHostFree(e->cidx_host[i]);
            if (e->cidx_dev[i]) (void)hipFree(e->cidx_dev[i]);
            e->cidx_host[i] = e->cidx_dev[i] = nullptr;
            if (hipHostMalloc((void **)&e->cidx_host[i], (size_t)batch * sizeof(long long)) != hipSuccess ||
                hipMalloc((void **)&e->cidx_dev[i], (size_t)batch * sizeof(long long)) != hipSuccess)
                return fail(AZX_ENOMEM, "allocating the collate index ring failed");
            if (!e->cidx_ev[i]) HIPCHECK(hipEventCreateWithFlags(&e->cidx_ev[i], hipEventDisableTiming));
        }
        if (!e->cidx_maxk) TRY(dev_alloc(e, &e->cidx_maxk, 4));
        e->cidx_cap = batch;
    }
    const int slot = (int)(e->cidx_next++ % 8);
    HIPCHECK(hipEventSynchronize(e->cidx_ev[slot]));          // returns at once for a slot never used
    memcpy(e->cidx_host[slot], indices, (size_t)batch * sizeof(int64_t));
    HIPCHECK(hipMemcpyAsync(e->cidx_dev[slot], e->cidx_host[slot], (size_t)batch * sizeof(int64_t), hipMemcpyHostToDevice, st));
    azx_launch_replay_collate(e->ring, e->cidx_dev[slot], (int)batch, e->d.ncells, (long long *)color_dev,
                              legal_moves_dev, (long long *)result_dev, board_dev, moves_prob_dev, reward_dev,
                              e->cidx_maxk, e->ring_mover_view ? e->d.N : 0, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipEventRecord(e->cidx_ev[slot], st));
    return AZX_OK;
}

// ---- float32 arithmetic self-test hook (tests): IEEE sqrt/divide and no FMA contraction ----
extern "C" int azx_selftest_arith(int device, int n, const float *a, const float *b, float *sq,
                                  float *dv, float *mul) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AZX_ENODEV, "no HIP device visible");
    DevGuard guard(device);
    float *da, *db, *ds, *dd, *dm;
    HIPCHECK(hipMalloc(&da, n * 4)); HIPCHECK(hipMalloc(&db, n * 4)); HIPCHECK(hipMalloc(&ds, n * 4));
    HIPCHECK(hipMalloc(&dd, n * 4)); HIPCHECK(hipMalloc(&dm, n * 4));
    HIPCHECK(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice));
    azx_launch_arith(da, db, ds, dd, dm, n, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(sq, ds, n * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(dv, dd, n * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(mul, dm, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(ds); (void)hipFree(dd); (void)hipFree(dm);
    return AZX_OK;
}

// ---- self-test hook (tests): the search kernel's unscaled divide and sqrt table ----------------
extern "C" int azx_selftest_divide(int device, int n, const float *num, const float *den, float *quot,
                                   float *sqrt_tab) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(AZX_ENODEV, "no HIP device visible");
    if (n < 1 || !num || !den || !quot || !sqrt_tab) return fail(AZX_EINVAL, "bad argument");
    DevGuard guard(device);
    if (azx_init_geometry(device)) return fail(AZX_EHIP, "uploading the constant tables failed");
    float *da, *db, *dq, *dr;
    HIPCHECK(hipMalloc(&da, n * 4)); HIPCHECK(hipMalloc(&db, n * 4));
    HIPCHECK(hipMalloc(&dq, n * 4)); HIPCHECK(hipMalloc(&dr, n * 4));
    HIPCHECK(hipMemcpy(da, num, n * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(db, den, n * 4, hipMemcpyHostToDevice));
    azx_launch_divide_test(da, db, dq, dr, n, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(quot, dq, n * 4, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(sqrt_tab, dr, n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dq); (void)hipFree(dr);
    return AZX_OK;
}

// ---- device Dirichlet self-test hook (tests): n_rows draws of Dirichlet(alpha * 1_k) -------------
extern "C" int azx_selftest_dirichlet(int device, double alpha, int k, int n_rows, uint32_t seed, float *out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(AZX_ENODEV, "no HIP device visible");
    if (k < 1 || k > 128 || n_rows < 1 || !out) return fail(AZX_EINVAL, "bad argument");
    DevGuard guard(device);
    float *d = nullptr;
    HIPCHECK(hipMalloc(&d, sizeof(float) * (size_t)k * n_rows));
    if (!(alpha > 0.0)) return fail(AZX_EINVAL, "alpha must be positive");
    std::vector<float> tab(AZX_GAMMA_TAB_FLOATS);
    azx_gamma_table(alpha, tab.data());
    float *dt = nullptr;
    HIPCHECK(hipMalloc(&dt, tab.size() * sizeof(float)));
    HIPCHECK(hipMemcpy(dt, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
    azx_launch_noise_test((float)alpha, dt, k, n_rows, seed, d, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, d, sizeof(float) * (size_t)k * n_rows, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    (void)hipFree(dt);
    return AZX_OK;
}

extern "C" int azx_debug_choose(azx_engine *e, int32_t *move_id, float *moves_prob) {
    if (!e || !move_id || !moves_prob) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    DevEngine &d = e->d;
    TRY(play_setup(e, std::max<int64_t>(e->q_alloc, 1 << 10), 1));
    std::vector<GameHdr> before(d.G), after(d.G);
    HIPCHECK(hipMemcpyAsync(before.data(), d.ghdr, sizeof(GameHdr) * d.G, hipMemcpyDeviceToHost, e->stream));
    azx_launch_choose(d, e->stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpyAsync(after.data(), d.ghdr, sizeof(GameHdr) * d.G, hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    std::vector<float> row(AZX_CELL_STRIDE);
    for (int g = 0; g < d.G; ++g) {
        const bool drew = after[g].n_rows == before[g].n_rows + 1;
        move_id[g] = drew ? after[g].move_id : -1;
        float *out = moves_prob + (size_t)g * d.ncells;
        if (!drew) { std::fill(out, out + d.ncells, 0.0f); continue; }
        HIPCHECK(hipMemcpyAsync(row.data(), d.row_prob + ((size_t)g * d.ncells + before[g].n_rows) * AZX_CELL_STRIDE,
                                sizeof(float) * AZX_CELL_STRIDE, hipMemcpyDeviceToHost, e->stream));
        HIPCHECK(hipStreamSynchronize(e->stream));
        std::copy(row.begin(), row.begin() + d.ncells, out);
    }
    return AZX_OK;
}

extern "C" int azx_debug_counters_raw(azx_engine *e, uint64_t *out, int64_t n_games) {
    if (!e || !out) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    if (n_games != e->d.G) return fail(AZX_EINVAL, "n_games must equal the engine's game count");
    HIPCHECK(hipMemcpyAsync(out, e->d.counters, (size_t)n_games * CTR_COUNT * sizeof(unsigned long long),
                            hipMemcpyDeviceToHost, e->stream));
    HIPCHECK(hipStreamSynchronize(e->stream));
    return AZX_OK;
}

extern "C" int azx_debug_counters(azx_engine *e, uint64_t *out16) {
    if (!e || !out16) return fail(AZX_EINVAL, "null argument");
    ENGINE_GUARD(e);
    CounterSnap snap;
    TRY(snap_counters(e, &snap));
    for (int j = 0; j < CTR_COUNT; ++j) out16[j] = snap.c[j];
    return AZX_OK;
}

// ---- native training step (train_kernels.hip) ----------------------------------------------------------------
struct azx_trainer {
    AzxTrain *t = nullptr;
    int device = 0;
};

#define TRAIN_GUARD(h) DevGuard _dev_guard((h)->device)
static int trn_rc(int rc) {
    if (rc) g_err = azx_trn_error();
    return rc;
}

extern "C" int azx_train_create(const azx_train_config *cfg, azx_trainer **out) {
    if (!cfg || !out) return fail(AZX_EINVAL, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(AZX_ENODEV, "no HIP device visible: the training step is HIP-only (no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(AZX_EINVAL, "device %d out of range (%d visible)", cfg->device, ndev);
    DevGuard guard(cfg->device);
    azx_trainer *h = new azx_trainer();
    h->device = cfg->device;
    int rc = azx_trn_create(&h->t, cfg->board_size, cfg->num_blocks, cfg->base_chans, cfg->batch_size, cfg->device);
    if (rc) { delete h; return trn_rc(rc); }
    *out = h;
    return AZX_OK;
}

extern "C" void azx_train_destroy(azx_trainer *h) {
    if (!h) return;
    { TRAIN_GUARD(h); azx_trn_destroy(h->t); }
    delete h;
}

extern "C" int azx_train_bind(azx_trainer *h, int n, const char *const *names, void *const *tensors,
                              const int64_t *counts, void *const *momentum) {
    if (!h || !names || !tensors || !counts) return fail(AZX_EINVAL, "null argument");
    TRAIN_GUARD(h);
    return trn_rc(azx_trn_bind(h->t, n, names, tensors, counts, momentum));
}

extern "C" int azx_train_inputs(azx_trainer *h, int32_t **board, int32_t **legal_moves, float **moves_prob, float **reward) {
    if (!h) return fail(AZX_EINVAL, "null argument");
    return trn_rc(azx_trn_inputs(h->t, board, legal_moves, moves_prob, reward));
}

extern "C" int azx_train_outputs(azx_trainer *h, float **loss3, float **value, float **moves_logprob) {
    if (!h) return fail(AZX_EINVAL, "null argument");
    return trn_rc(azx_trn_outputs(h->t, loss3, value, moves_logprob));
}

extern "C" int azx_train_step(azx_trainer *h, float lr, float momentum, float weight_decay, void *hip_stream) {
    if (!h) return fail(AZX_EINVAL, "null argument");
    TRAIN_GUARD(h);
    return trn_rc(azx_trn_step(h->t, lr, momentum, weight_decay, (hipStream_t)hip_stream));
}

extern "C" int azx_train_debug(azx_trainer *h, const char *name, void *out, int64_t cap, int64_t *nbytes) {
    if (!h || !name || !nbytes) return fail(AZX_EINVAL, "null argument");
    TRAIN_GUARD(h);
    return trn_rc(azx_trn_debug(h->t, name, out, cap, nbytes));
}
