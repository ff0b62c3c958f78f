// mcts_kernels.hip -- Hex movegen + flat-array PUCT search kernels for gfx950 (MI355X).
//
// One 64-lane wavefront owns one concurrent game.  The root board lives in registers (lane l
// holds cells l, l+64[, l+128]; its stones also as scalar bitboards), the root's children and a
// 64-entry cache of deeper nodes stay on chip for a whole launch, the children of a deeper node
// are read with one coalesced 16-byte load per lane, the PUCT argmax is a wave reduction, a
// descent only tracks occupied cells and the Hex win test runs at new leaves on the root's group
// labels (O(1) wave-parallel relabel, HexWave in azx_dev.h).  DESIGN.md section 3.1 has the rest.
//
// Restates (bit-exact in float32, compile with -ffp-contract=off):
//   azalea/mcts.py:46-76 select_batch, :79-92 apply_virtual_loss, :95-116 select_leaf,
//   :119-136 score_actions, :139-152 deduplicate_leaves, :155-217 evaluate_batch (host half),
//   :226-239 expand_batch, :242-255 backup_batch, :258-293 sample_paths;
//   azalea/search_tree.py:59-71 reset, :115-132 move, :254-274 create_child_nodes;
//   azalea/game/hex.py:137-231 rules, :72-122 perspective flip.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "azx_dev.h"
#include "mcts_kernels.h"

__constant__ uint64_t c_geo[AZX_GEO_CELLS * 4];
__constant__ float c_sqrt[AZX_SQRT_TAB];
__constant__ float c_invk[256];       // float32 1/k, the default uniform prior table (same bits as the IEEE divide)

// fill c_geo for every supported board size (once per device)
int azx_init_geometry(int device) {
    static bool done[64] = {false};
    if (device >= 0 && device < 64 && done[device]) return 0;
    static uint64_t tab[AZX_GEO_CELLS * 4];
    memset(tab, 0, sizeof tab);
    for (int N = 2; N <= AZX_MAX_BOARD; ++N) {
        const int base = azx_geo_base(N);
        const int dr[6] = {-1, -1, 0, 0, 1, 1}, dc[6] = {0, 1, -1, 1, -1, 0};   // hex.py:190-195
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) {
                uint64_t *g = tab + (size_t)(base + r * N + c) * 4;
                for (int d = 0; d < 6; ++d) {
                    const int rr = r + dr[d], cc = c + dc[d];
                    if (rr >= 0 && rr < N && cc >= 0 && cc < N) {
                        const int cell = rr * N + cc;
                        g[cell >> 6] |= 1ull << (cell & 63);
                    }
                }
                g[3] = (r == 0 ? 1ull : 0ull) | (r == N - 1 ? 2ull : 0ull) |
                       (c == 0 ? 4ull : 0ull) | (c == N - 1 ? 8ull : 0ull);
            }
    }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_geo), tab, sizeof tab) != hipSuccess) return -1;
    static float rt[AZX_SQRT_TAB];
    for (int i = 0; i < AZX_SQRT_TAB; ++i) rt[i] = sqrtf((float)i);   // IEEE: same bits as np.sqrt(float32)
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_sqrt), rt, sizeof rt) != hipSuccess) return -1;
    static float ik[256];
    ik[0] = 0.0f;
    for (int i = 1; i < 256; ++i) ik[i] = 1.0f / (float)i;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_invk), ik, sizeof ik) != hipSuccess) return -1;
    if (device >= 0 && device < 64) done[device] = true;
    return 0;
}

// Diagnostic build only (-DAZX_STAMP): per-region s_memtime sums in counters[10..15].  The
// shipped kernel executes no stamp.
#ifdef AZX_STAMP
#define T_DECL unsigned long long t_last = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0};
#define T_MARK(r) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[r] += t_now - t_last; t_last = t_now; }
#if AZX_STAMP == 2   // load-balance diagnostic: slot 10 = this search's wave lifetime, 11 = HW_ID, 12 = start time, 13 = lifetimes summed
#define T_FLUSH if (lane == 0) { unsigned long long *c_ = E.counters + (size_t)g * CTR_COUNT; \
    c_[13] += __builtin_amdgcn_s_memtime() - t_start; c_[10] = __builtin_amdgcn_s_memtime() - t_start; c_[11] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32); c_[12] = t_start; }
#else
#define T_FLUSH if (lane == 0) { for (int r_ = 0; r_ < 6; ++r_) E.counters[(size_t)g * CTR_COUNT + 10 + r_] += t_acc[r_]; }
#endif
#else
#define T_DECL
#define T_MARK(r)
#define T_FLUSH
#endif

__device__ __forceinline__ int popc64(uint64_t x) { return __popcll(x); }
// bit `lane` of a wave-uniform mask as a lane predicate (v_cndmask on the SGPR pair) and the
// number of set bits below this lane (v_mbcnt): no 64-bit vector shifts
__device__ __forceinline__ bool lane_bit(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ int rank_below(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// ---- Philox4x32-10 counter RNG (per-game stream key; the device move draw in k_choose) -------
struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t out[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};
__device__ __forceinline__ float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
// global index of the gen-th game slot g starts (seeds come from the global game index, so
// a fixed seed plays the same set of games on 1, 2 or 8 GPUs -- rank r of W passes stride W, offset r)
__device__ __forceinline__ int64_t game_uid(const DevEngine &E, int g, int gen) {
    return ((int64_t)g + (int64_t)E.G * gen) * E.uid_stride + E.uid_offset;
}
__device__ __forceinline__ Philox game_rng(const DevEngine &E, int64_t uid) {
    Philox ph;
    const uint64_t s = E.seed + (uint64_t)uid;
    ph.k0 = (uint32_t)s;
    ph.k1 = (uint32_t)(s >> 32) ^ 0x5bd1e995u;
    return ph;
}

// 32-bit mixer (two multiplies per word instead of Philox's forty per four): the Dirichlet
// noise only needs decorrelated uniforms, it is re-drawn 410 times per move per cell
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// order-preserving float32 <-> uint32 map (total order of the non-NaN floats): wave maxima of
// floats become one v_max_u32 per DPP step, and the inverse runs on the scalar unit
__device__ __forceinline__ uint32_t f32_key(float x) {
    const uint32_t b = (uint32_t)__float_as_int(x);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) {
    return __int_as_float((int)((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k));
}

// log2 of a Gamma(alpha) variate, 0 < alpha < 1, by inversion of the CDF -- no rejection loop (a
// loop runs until the unluckiest of 64 lanes accepts).  Only ratios of the variates matter to the
// Dirichlet, and for alpha = 0.03 x = u^(1/alpha) underflows float32 all the time, so everything
// stays in log2:
//     log2 x = (log2 u + log2 Gamma(alpha+1)) / alpha + delta(1 - u)
// The first term is the inverse of the small-x form F(x) = x^alpha / Gamma(alpha+1) and exact to
// float32 while x < 2^-24 (u < 0.6 for alpha = 0.03); delta, the smooth remainder, is tabulated by
// the host in double precision (azx_gamma_table) on a grid that follows the float format of
// y = 1 - u -- 32 points per octave from 2^-25 to 1, indexed by the exponent and top mantissa bits
// of y, no second logarithm -- and interpolated linearly (|error| < 1e-4 in log2 x at the far
// tail, far less elsewhere).  u and y come from the same 24 random bits, each exact in float32.
#define AZX_GAMMA_TAB 801            // 25 octaves x 32 + 1
struct GammaConst { float inv_alpha, c0; const float *tab; };
__device__ __forceinline__ GammaConst gamma_const(float alpha, const float *tab) {
    GammaConst g;
    g.inv_alpha = 1.0f / alpha;
    g.c0 = tab[AZX_GAMMA_TAB];       // log2 Gamma(alpha + 1), appended by the host
    g.tab = tab;
    return g;
}
__device__ __forceinline__ float log2_gamma_variate(uint32_t seed, const GammaConst &gc) {
    const uint32_t r = mix32(seed) >> 8;                                      // 24 random bits
    const float u = __builtin_fmaf((float)r, 1.0f / 16777216.0f, 0.5f / 16777216.0f);
    const float y = __builtin_fmaf((float)(r ^ 0xffffffu), 1.0f / 16777216.0f, 0.5f / 16777216.0f);   // 1 - u
    const float lx0 = (__builtin_amdgcn_logf(u) + gc.c0) * gc.inv_alpha;
    const uint32_t yb = (uint32_t)__float_as_int(y);
    const int idx = (int)(yb >> 18) - ((127 - 25) << 5);                      // 0 .. 799
    const float fr = (float)(yb & 0x3ffffu) * (1.0f / 262144.0f);
    const float t0 = gc.tab[idx], t1 = gc.tab[idx + 1];
    return lx0 + __builtin_fmaf(fr, t1 - t0, t0);
}

// scale * Dirichlet(alpha * 1_k) over the cells set in m[] (one value per lane-slot), the device
// counterpart of rng.dirichlet(np.full(k, alpha)) at mcts.py:128: independent Gamma(alpha)
// variates normalised by their sum, computed from log2-variates with the maximum subtracted.
template <int SLOTS>
__device__ __forceinline__ void dirichlet_noise(const uint64_t *m, int lane, uint32_t stream,
                                                const GammaConst &gc, float scale, float *nz) {
    // every lane draws (the table index is valid for any 24 random bits), cells that are not legal
    // are masked afterwards: no divergent region around the two table lookups, both in flight together
    float lg[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
        lg[s] = log2_gamma_variate(stream ^ ((uint32_t)(s * 64 + lane) * 0x9E3779B9u), gc);
    // The variates only matter relative to each other.  The largest of k of them is almost always
    // within 2^-60 .. 2^4 (it is below 2^-60 when every u < 0.29: 0.29^k), so they are first summed
    // as they are; only when that sum underflows (late in a game, k of a few cells) is the maximum
    // taken out in log space first.  Either way the result is w_i / sum(w) to float32 rounding.
    float sw = 0.0f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        nz[s] = lane_bit(m[s]) ? __builtin_amdgcn_exp2f(lg[s]) : 0.0f;
        sw += nz[s];
    }
    float tot = wave_sum(sw);
#ifdef AZX_NOISE_MAXFIRST   // diagnostic build: always take the rescaling path
    if (true) {
#else
    if (!(tot >= 8.67361738e-19f)) {
#endif                  // < 2^-60 (or no finite value at all): rescale
        uint32_t kmax = 0u;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const uint32_t k = lane_bit(m[s]) ? f32_key(lg[s]) : 0u;
            kmax = k > kmax ? k : kmax;
        }
        const float mx = key_f32(wave_max_u32(kmax));   // >= one cell is set whenever this is called
        sw = 0.0f;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            nz[s] = lane_bit(m[s]) ? __builtin_amdgcn_exp2f(lg[s] - mx) : 0.0f;
            sw += nz[s];
        }
        tot = wave_sum(sw);
    }
    sw = scale * __builtin_amdgcn_rcpf(tot);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) nz[s] = nz[s] * sw;
}

template <int SLOTS>
struct Masks {
    uint64_t m[SLOTS];
    int base[SLOTS];
    int k;
};

template <int SLOTS>
__device__ __forceinline__ Masks<SLOTS> make_masks(const HexWave<SLOTS> &h, int lane, int ncells) {
    Masks<SLOTS> mk;
    int k = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        mk.m[s] = h.empties(s, lane, ncells);
        mk.base[s] = k;
        k += popc64(mk.m[s]);
    }
    mk.k = h.winner ? 0 : k;   // hex.py:152-153: no legal moves once there is a winner
    return mk;
}

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

__device__ __forceinline__ void wave_mem_sync() {
    // global stores of this wave must be visible to later loads issued by OTHER lanes of it
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// source cell of output cell o after the perspective flip (hex.py:72-87)
__device__ __forceinline__ int flip_src(int o, int N) {
    const int i = o / N, j = o - i * N;
    return (N - 1 - j) * N + (N - 1 - i);
}

struct Lds {
    int32_t *path;         // [bs][pstride] node ids along each selected path (index d = depth d+1)
    unsigned char *colors; // [bs][AZX_CELL_STRIDE] absolute colours at the leaf (only when leaf boards leave the wave)
    float *gtab;           // [AZX_GAMMA_TAB + 1] the gamma sampler's table, staged once per launch
};

size_t azx_mcts_lds_bytes(int ncells, int bs) {
    return (size_t)bs * (ncells + (ncells & 1)) * 4 + (size_t)bs * AZX_CELL_STRIDE + 64 +
           (size_t)(AZX_GAMMA_TAB + 1) * 4;
}

__device__ __forceinline__ Lds carve_lds(unsigned char *raw, int ncells, int bs) {
    Lds L;
    L.path = reinterpret_cast<int32_t *>(raw);
    L.colors = reinterpret_cast<unsigned char *>(L.path + bs * (ncells + (ncells & 1)));
    L.gtab = reinterpret_cast<float *>(L.colors + bs * AZX_CELL_STRIDE + 64);   // bs*192 and the rest are multiples of 8
    return L;
}

__device__ __forceinline__ float readlane_f(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// ---- score arithmetic, two board slots per instruction (v_pk_mul/add/fma_f32) -----------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
// Correctly rounded float32 n/d: the reciprocal-refine-correct FMA chain the compiler emits for
// an IEEE '/', without its v_div_scale / v_div_fixup range handling.  Exact (same bits as '/')
// whenever d is normal in [1, 2^24] and n is 0 or 2^-100 <= |n| <= 2^100: no intermediate
// underflows, so scaling would be the identity.  (A -0 numerator gives +0; the score adds 0.0
// afterwards, mcts.py:135, so the sign of a zero never reaches the argmax.)
__device__ __forceinline__ f2 div2_unscaled(f2 n, f2 d) {
    f2 r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    const f2 one = {1.0f, 1.0f};
    f2 e = fma2(-d, r, one);
    r = fma2(e, r, r);
    f2 q = n * r;
    e = fma2(-d, q, n);
    q = fma2(e, r, q);
    e = fma2(-d, q, n);
    return fma2(e, r, q);
}
__device__ __forceinline__ f2 div2_ieee(f2 n, f2 d) {
    f2 q;
    q.x = n.x / d.x;
    q.y = n.y / d.y;
    return q;
}
// values whose backup keeps every total_value either 0 or >= 2^-83 in magnitude (sums of
// multiples of 2^-83 stay multiples of it): 0 or 2^-60 <= |v|
__device__ __forceinline__ bool value_in_fast_range(float v) {
    const float a = __builtin_fabsf(v);
    return v == 0.0f || (a >= 8.67361738e-19f && a <= 1.0e30f);
}
// copy of a wave-uniform value into a vector register AT THIS POINT of the program: the compiler
// otherwise copies a scalar load's result where it is defined and waits for the load there
__device__ __forceinline__ float late_vgpr(float s) {
    float v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
    return v;
}
// wave-uniform float load on the scalar unit (lgkmcnt): keeps vmcnt free of read-after-store waits
__device__ __forceinline__ float sload_f32(const float *p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// ============================================================================================
// The search kernel.  mode = MODE_* bits (mcts_kernels.h).
//
// On-chip tree state (what makes the kernel one HBM round trip per simulation instead of six):
//  * the root's children (stats + link) live in registers, lane = board cell, for the whole
//    launch: every descent's first level, and every virtual-loss / backup update of a depth-1
//    node, is pure ALU;
//  * deeper nodes touched by a path sit in a 64-entry write-back cache (lane j = entry j:
//    id, num_visits, total_value).  They enter it with the values just loaded on the way down,
//    virtual loss / undo / backup update the entry, child-block loads are patched from it, and
//    it is written back when full and at the end of the launch.
// Every update is still the same sequence of float32 additions per node as mcts.py:79-92 /
// :242-255, so results are bit-identical to the sequential reference.
// ============================================================================================
// FAST = the throughput path of BASELINE configs[1] as its own instantiation: one launch per move
// (MODE_BEGIN | MODE_INLINE), AZX_EVAL_UNIFORM with the default 1/k prior table, noise off or drawn
// on the device.  The same source with those conditions folded at compile time: the evaluator
// hand-off, host-noise and slow-divide paths disappear and with them the scalar registers that held
// their pointers (the generic kernel spills ~40 SGPRs inside the select loop).  Results are
// bit-identical to the generic instantiation (test_fast_kernel_matches_generic).
// ============================================================================================
// Arena compaction: Cheney copy of the subtree under `child` into the other arena, level by level
// (children of a level-L node: kL of them; the kept subtree keeps the reference's breadth-first
// child order, search_tree.py:254-274).  `sh` = 64 ints of LDS.
// ============================================================================================
__device__ __forceinline__ void compact_tree(const DevEngine &E, int g, TreeHdr *th, int child, int k_child,
                                             int lane, int *sh_old) {
    const Node *src = E.arena[th->arena] + (size_t)g * E.cap;
    Node *dst = E.arena[th->arena ^ 1] + (size_t)g * E.cap;
    if (lane == 0) dst[0] = src[child];
    wave_mem_sync();
    int n_new = 1, lvl_start = 0, lvl_end = 1, kL = k_child;
    while (lvl_start < lvl_end && kL > 0) {
        for (int i0 = lvl_start; i0 < lvl_end; i0 += 64) {
            const int i = i0 + lane;
            int oldfc = -1;
            if (i < lvl_end) oldfc = dst[i].link;
            const bool has = oldfc >= 0;
            const uint64_t hm = __ballot(has);
            const int newfc = n_new + kL * rank_below(hm);
            if (has) dst[i].link = newfc;
            // the child blocks of this group's parents land back to back at n_new: one flat copy over
            // (parent, child) pairs, four 16-byte loads in flight per lane
            const int np = popc64(hm);
            if (has) sh_old[rank_below(hm)] = oldfc;
            lds_sync();
            const int total = kL * np;
            const float inv_k = 1.0f / (float)kL;
            for (int t0 = 0; t0 < total; t0 += 256) {
                Node v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * 64 + lane;
                    if (t < total) {
                        const int pp = (int)(((float)t + 0.5f) * inv_k);   // t / kL (exact: t < 2^14)
                        v[u] = src[sh_old[pp] + (t - pp * kL)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * 64 + lane;
                    if (t < total) dst[n_new + t] = v[u];
                }
            }
            lds_sync();
            n_new += total;
            wave_mem_sync();
        }
        lvl_start = lvl_end;
        lvl_end = n_new;
        kL -= 1;
    }
    if (lane == 0) {
        th->arena ^= 1;
        th->dropped += th->num_nodes - n_new;
        th->num_nodes = n_new;
        th->root_id = 0;
        th->root_k = k_child;
        th->k0 = k_child;
        th->defer_compact = 0;
    }
}

#ifdef AZX_WPE
#define AZX_MCTS_ATTR __attribute__((amdgpu_waves_per_eu(AZX_WPE, AZX_WPE)))
#else
#define AZX_MCTS_ATTR
#endif
template <int SLOTS, bool FAST>
__device__ __forceinline__ void mcts_body(const DevEngine &E, int mode_arg, int num_batches, bool stage_tables = true) {
    const int mode = FAST ? (MODE_BEGIN | MODE_INLINE) : mode_arg;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    const int N = E.N, ncells = E.ncells, bs = E.bs;
    const int pstride = ncells + (ncells & 1);
    GameHdr *gh = E.ghdr + g;
    if (!gh->active) return;
#ifdef AZX_STAMP
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    TreeHdr *th = E.thdr + g;
    const Lds L = carve_lds(smem_raw, ncells, bs);
    if ((mode & MODE_BEGIN) && th->defer_compact) {
        // the arena compaction k_advance left to this launch (play mode): this game's copy runs
        // beside the other games' searches
        compact_tree(E, g, th, th->root_id, th->root_k, lane, reinterpret_cast<int *>(L.path));
        __builtin_amdgcn_s_waitcnt(0);   // the header fields written by lane 0 are re-read below
        wave_mem_sync();
    }

    int num_nodes = th->num_nodes;
    const int root_id = th->root_id;
    int status = th->status;
    int batches_left = th->batches_left;
    int pending = th->pending;
    int pending_root = th->pending_root;
    int select_count = th->select_count;
    float search_value = th->search_value;
    bool slow_div = FAST ? false : th->slow_div != 0;   // FAST: values are 0 or -1 only
    Node *arena = E.arena[th->arena] + (size_t)g * E.cap;

    HexWave<SLOTS> root;
    root.load(E.cells + (size_t)g * SLOTS * 64, lane);
    root.geom(N, lane);
    root.color = gh->color;
    root.winner = gh->winner;
    const int ply = gh->ply;
    const Masks<SLOTS> rootmk = make_masks<SLOTS>(root, lane, ncells);

    // per-launch tallies (at most 16 * 410 * 169 < 2^32 each): 32-bit scalars
    uint32_t c_selects = 0, c_depth = 0, c_kint = 0, c_kleaf = 0, c_evals = 0, c_term = 0;
    const bool inline_eval = FAST || (mode & MODE_INLINE) != 0;
    const bool need_colors = FAST ? false : (!inline_eval || E.evaluator == AZX_EVAL_UNIFORM_HASH);   // leaf boards leave the wave
    const float c32 = E.c_puct;
    const float keep32 = (float)(1.0 - E.noise_scale);   // python float -> f32 (weak scalar)
    const Philox ph = game_rng(E, gh->uid);
    // the device sampler's table goes to LDS (two dependent lookups per cell and select)
    // (k_play stages it once per LAUNCH: the area is this table's alone, nothing between two moves writes it)
    if (stage_tables && E.noise_scale != 0.0 && (FAST || E.device_noise)) {
        for (int i = lane; i < AZX_GAMMA_TAB + 1; i += 64) L.gtab[i] = E.gamma_tab[i];
        lds_sync();
    }
    const GammaConst gconst = gamma_const(E.noise_alpha, L.gtab);
    const uint32_t noise_base = mix32(ph.k0 ^ mix32(ph.k1 + (uint32_t)ply * 0x632be5abu));

    if (mode & MODE_BEGIN) {
        batches_left = num_batches;
        select_count = 0;
        search_value = 0.0f;
        pending = 0;
        pending_root = 0;
    }

    // ---- on-chip tree state ----------------------------------------------------------------
    // Root children live in registers BY RANK (lane-slot j = child j, the layout they have in HBM):
    // once 64 or fewer moves remain the whole root level -- noise, scores, argmax -- runs on one
    // register slot.  rcell = the board cell of that child.
    float4 rst[SLOTS];                 // root children: {num_visits, total_value, prior, link}
    int rcell[SLOTS];
    bool rempty[SLOTS];
    const int k_root = rootmk.k;
    uint64_t rootrm[SLOTS];            // lane masks of the root's children: ranks 64 s .. of k_root
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int n = k_root - 64 * s;
        rootrm[s] = n >= 64 ? ~0ull : (n <= 0 ? 0ull : ((1ull << n) - 1ull));
    }
    // Winning moves at the root, per colour: an empty cell whose own edge flags OR-ed with the flags of
    // the same-coloured groups around it reach both edges (hex.py:204-231 for one new stone).  A leaf
    // at depth <= 2 has exactly one stone of the last mover on its path, so its win test is one bit
    // of these masks -- computed once per launch instead of at every new leaf.
    uint64_t winm[2][SLOTS];
    {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s * 64 + lane < ncells) L.path[s * 64 + lane] = (int32_t)root.c[s];
        lds_sync();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int cell = s * 64 + lane;
            uint32_t f1 = 0u, f2 = 0u;
            const bool empty = cell < ncells && (root.c[s] & 3u) == 0u;
            if (empty) {
                const uint64_t *gq = c_geo + (size_t)(root.gbase + cell) * 4;
                const uint32_t edge = (uint32_t)gq[3];
                f1 = edge & 3u;
                f2 = (edge >> 2) & 3u;
#pragma unroll
                for (int t = 0; t < SLOTS; ++t) {
                    uint64_t m = gq[t];
                    while (m) {
                        const int j = (int)__ffsll((long long)m) - 1;
                        m &= m - 1;
                        const uint32_t w = (uint32_t)L.path[t * 64 + j];
                        const uint32_t fl = (w >> 2) & 3u;
                        if ((w & 3u) == 1u) f1 |= fl;
                        if ((w & 3u) == 2u) f2 |= fl;
                    }
                }
            }
            winm[0][s] = __ballot(empty && f1 == 3u);
            winm[1][s] = __ballot(empty && f2 == 3u);
        }
        lds_sync();
    }
    {   // rank -> cell through LDS (the path area is free until the first descent)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (lane_bit(rootmk.m[s])) L.path[rootmk.base[s] + rank_below(rootmk.m[s])] = s * 64 + lane;
        lds_sync();
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) rcell[s] = (s * 64 + lane < k_root) ? L.path[s * 64 + lane] : 0;
        lds_sync();
    }
    float root_nv, root_tv;
    int root_link;
    {
        const Node rn = arena[root_id];
        root_nv = rn.nv;
        root_tv = rn.tv;
        root_link = rn.link;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            rempty[s] = s * 64 + lane < k_root;
            rst[s] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (root_link >= 0 && rempty[s])
                rst[s] = *reinterpret_cast<const float4 *>(arena + root_link + s * 64 + lane);
        }
    }
    float rkp[SLOTS];                  // (1 - eps) * prior of the root children
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) rkp[s] = keep32 * rst[s].z;
    // sum of the root children's visit counts (the integer under the square root of mcts.py:132),
    // kept as a scalar: every change of a root child's count goes through root_child_add
    int root_sumn = 0;
    {
        int t = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) t += (int)rst[s].x;
        root_sumn = wave_sum_i(t);
    }
    // Everything loaded so far is waited for HERE, once per launch.  Without it the compiler, unable to
    // prove that root_nv / root_tv had arrived on every path into the batch loop, put a full
    // s_waitcnt vmcnt(0) in front of their update at every backed-up leaf -- which also waits for the
    // child blocks the previous leaf's expansion had just stored (an HBM write round trip per leaf).
    float invk_reg = c_invk[(k_root - lane) & 255];   // see inline_prior
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    asm volatile("" : "+v"(root_nv), "+v"(root_tv), "+v"(invk_reg));
    uint64_t rootall[SLOTS];           // cells that are not empty at the root: stones + off-board bits
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int nvalid = ncells - 64 * s;
        const uint64_t valid = nvalid >= 64 ? ~0ull : (nvalid <= 0 ? 0ull : ((1ull << nvalid) - 1ull));
        rootall[s] = root.occ[0][s] | root.occ[1][s] | ~valid;
    }
    int c_id = -1, n_c = 0;            // node cache: entry j lives in lane j
    float c_nv = 0.0f, c_tv = 0.0f;
    // per-leaf metadata of the batch in flight: leaf i lives in lane i (v_readlane to fetch)
    int m_node = -1, m_len = 0, m_link = 0, m_tm = 0, m_cells = 0, m_uidx = 0, m_k = 0, m_slot1 = -1;
    float m_val = 0.0f;
    uint64_t m_mask[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) m_mask[s] = 0ull;
    auto rl = [&](int v, int i) -> int { return __builtin_amdgcn_readlane(v, i); };
    auto rl64 = [&](uint64_t v, int i) -> uint64_t {
        return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), i) << 32) |
               (uint32_t)__builtin_amdgcn_readlane((int)v, i);
    };
    // numpy's pairwise float32 sum over the first n lanes' m_val (mcts.py:287)
    auto np_sum_vals = [&](int n) -> float {
        float a[AZX_MAX_BATCH];
#pragma unroll
        for (int i = 0; i < AZX_MAX_BATCH; ++i) a[i] = readlane_f(m_val, i);
        if (n < 8) {
            float r = 0.0f;
#pragma unroll
            for (int i = 0; i < 7; ++i) if (i < n) r += a[i];
            return r;
        }
        float r8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r8[j] = a[j];
        if (n >= 16) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r8[j] += a[8 + j];
        }
        float res = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
        const int done = n >= 16 ? 16 : 8;
#pragma unroll
        for (int i = 8; i < AZX_MAX_BATCH; ++i) if (i >= done && i < n) res += a[i];
        return res;
    };

    bool memo_ok = true;               // no flush since the batch began: remembered entry indices hold
    auto cache_flush = [&]() {
        if (lane < n_c) *reinterpret_cast<float2 *>(arena + c_id) = make_float2(c_nv, c_tv);
        n_c = 0;
        memo_ok = false;
        wave_mem_sync();
    };
    // entry index of node `id`; on a miss it is inserted with (nv0, tv0) when `known`
    // (values just read on the way down) or with the values in HBM
    auto cache_find = [&](int id, bool known, float nv0, float tv0) -> int {
        const uint64_t hit = __ballot(lane < n_c && c_id == id);
        if (hit) return (int)__ffsll((long long)hit) - 1;
        if (n_c == 64) cache_flush();
        if (!known) {
            const float2 x = *reinterpret_cast<const float2 *>(arena + id);
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) here, so the (usual) hit path never waits
            nv0 = x.x;
            tv0 = x.y;
        }
        if (lane == n_c) { c_id = id; c_nv = nv0; c_tv = tv0; }
        return n_c++;
    };
    auto root_child_add = [&](int rank, float dv, float dt) {
        const int ln = rank & 63, sl = rank >> 6;
        root_sumn += (int)dv;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s == sl && lane == ln) { rst[s].x += dv; rst[s].y += dt; }
    };
    // (num_visits += dv, total_value += +-amount) along one recorded path.  pth[d] is the node at
    // depth d+1, cell0 the RANK of the path's root child; the value added at the LEAF is `amount`, with `alternate` its sign flips at
    // every step towards the root (mcts.py:252); `with_root` includes the root (mcts.py:253).
    // slot1: cache entry of the depth-2 node as found during the descent (-1 = unknown): with it a
    // path of length <= 2 -- most of them -- is updated without reading the path or searching the cache
    auto path_apply = [&](const int32_t *pth, int len, int cell0, int slot1, float dv, float amount,
                          bool alternate, bool with_root) {
        if (with_root) {
            root_nv += dv;
            root_tv += (alternate && (len & 1)) ? -amount : amount;
        }
        if (len >= 1 && cell0 >= 0) root_child_add(cell0, dv, (alternate && ((len - 1) & 1)) ? -amount : amount);
        for (int d = 1; d < len; ++d) {
            const float a = (alternate && ((len - 1 - d) & 1)) ? -amount : amount;
            const int slot = (d == 1 && slot1 >= 0 && memo_ok) ? slot1
                                                               : cache_find(pth[d] & 0xffffff, false, 0.f, 0.f);
            if (lane == slot) { c_nv += dv; c_tv += a; }
        }
    };

    // ---- create_child_nodes (search_tree.py:254-274) for one leaf; returns false when full.
    // `cells` = rank of the path's root child | last cell << 16 (owner lanes of the link word).
    // With one prior for every child (prior_row == nullptr: the uniform evaluator) the block is written
    // by rank -- lane j stores child j, no masks needed (lm may be null, k_in = number of children);
    // those children are later loaded by other lanes (lane = cell), so the caller fences once per batch.
    auto expand = [&](int node, int len, int cells, int lnk, bool terminal, const uint64_t *lm, int k_in,
                      const float *prior_row, float prior_const) -> bool {
        if (lnk != AZX_LINK_UNEVAL) return true;          // re-selected terminal: mcts.py:237
        int k = 0;
        int base[SLOTS];
        if (lm) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) { base[s] = k; k += popc64(lm[s]); }
        } else {
            k = k_in;
        }
        if (terminal) k = 0;
        if (num_nodes + k > E.cap) { status = 1; return false; }   // SearchTreeFull
        const int fc = num_nodes;
        num_nodes += k;
        if (k > 0) {
            if (!prior_row) {
                Node nd;
                nd.nv = 0.0f;
                nd.tv = 0.0f;
                nd.pp = prior_const;
                nd.link = AZX_LINK_UNEVAL;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (s * 64 + lane < k) arena[fc + s * 64 + lane] = nd;
            } else
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                if (lane_bit(lm[s])) {
                    const int cell = s * 64 + lane;
                    const int rk = base[s] + rank_below(lm[s]);
                    Node nd;
                    nd.nv = 0.0f;
                    nd.tv = 0.0f;
                    nd.pp = prior_row ? prior_row[cell] : prior_const;
                    nd.link = AZX_LINK_UNEVAL;
                    arena[fc + rk] = nd;
                }
            }
            if (len == 0) {                               // the root's new children, by rank
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (rempty[s]) {
                        rst[s] = make_float4(0.f, 0.f, prior_row ? prior_row[rcell[s]] : prior_const,
                                             __int_as_float(AZX_LINK_UNEVAL));
                        rkp[s] = keep32 * rst[s].z;
                    }
            }
            c_kleaf += (uint32_t)k;
        }
        const int newlink = (k > 0) ? fc : AZX_LINK_TERM(fc);
        if (len == 0) {                                   // the root itself
            root_link = newlink;
            root_sumn = 0;
            if (lane == 0) arena[root_id].link = newlink;
        } else if (len == 1) {                            // a root child (by rank): its link is in registers
            const int c0 = cells & 0xffff, ln = c0 & 63, sl = c0 >> 6;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == sl && lane == ln) rst[s].w = __int_as_float(newlink);
        } else {
            // stored by the lane that will load this node as a child (same-lane program order)
            const int cl = (cells >> 16) & 0xffff;
            if (lane == (cl & 63)) arena[node].link = newlink;
        }
        return true;
    };

    // ---- emit one evaluation request: network-input board (flipped for O) ----------------
    auto emit_request = [&](int e, int src_index, const unsigned char *colors, int mover) {
        const bool flip = mover == 2;                       // state.color == 1: mcts.py:178
        for (int o = lane; o < AZX_CELL_STRIDE; o += 64) {
            unsigned char v = 0;
            if (o < ncells) {
                const int src = flip ? flip_src(o, N) : o;
                v = colors[src];
                if (flip && v) v = 3 - v;
            }
            E.ev_board[(size_t)e * AZX_CELL_STRIDE + o] = v;
        }
        if (lane == 0) {
            E.ev_src[e] = src_index;
            E.ev_flip[e] = flip ? 1 : 0;
        }
    };

    // fnv1a of the network-input board (parity stub value, tests/golden/make_golden.py)
    auto hash_value = [&](const unsigned char *colors, int mover) -> float {
        const bool flip = mover == 2;
        uint32_t h = 0x811c9dc5u;
        // each int32 contributes its low byte then three zero bytes: h = (h ^ b) * p^4
        const uint32_t p = 0x01000193u, p4 = p * p * p * p;
        for (int o = 0; o < ncells; ++o) {
            const int src = flip ? flip_src(o, N) : o;
            uint32_t v = colors[src];
            if (flip && v) v = 3 - v;
            h = (h ^ v) * p4;
        }
        return (float)((double)(h & 0xffffu) / 32768.0 - 1.0);
    };

    // A Hex position `len` plies below the root has exactly k_root - len legal moves, so the priors
    // a search can need are 1/(k_root - j): lane j keeps that entry of the float32 1/k table (one
    // vector load per launch) and an expansion reads it with v_readlane -- no memory access, where a
    // scalar load per expansion cost its latency every time.
    auto inline_prior = [&](int k, int len) -> float {
        if (FAST || E.prior_default) {
            if ((unsigned)len < 64u && k == k_root - len) return readlane_f(invk_reg, len);
            return c_invk[k & 255];
        }
        return sload_f32(E.prior_by_k + k);
    };

    // =================================== BEGIN: root evaluation (mcts.py:272-273) ========
    if ((mode & MODE_BEGIN) && status == 0 && root_link == AZX_LINK_UNEVAL) {
        if (!FAST) {
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int cell = s * 64 + lane;
                if (cell < ncells) L.colors[cell] = (unsigned char)(root.c[s] & 3u);
            }
            lds_sync();
        }
        if (inline_eval) {
            expand(root_id, 0, 0, AZX_LINK_UNEVAL, root.winner != 0, nullptr, rootmk.k, nullptr,
                   rootmk.k ? inline_prior(rootmk.k, 0) : 0.0f);
            wave_mem_sync();
            c_evals += 1;
        } else {
            int e = 0;
            if (lane == 0) e = atomicAdd(E.n_eval, 1);
            e = __builtin_amdgcn_readfirstlane(e);
            emit_request(e, g * bs + 0, L.colors, root.color);
            if (lane == 0) {
                const size_t lb = (size_t)g * bs;
                E.leaf_node[lb] = root_id;
                E.leaf_len[lb] = 0;
                E.leaf_eval[lb] = e;
                E.leaf_link[lb] = AZX_LINK_UNEVAL;
                E.leaf_cells[lb] = 0;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) E.leaf_mask[lb * 4 + s] = rootmk.m[s];
            }
            pending = 1;
            pending_root = 1;
            c_evals += 1;
        }
    }

    // =================================== APPLY: expand + backup pending leaves ============
    if (!FAST && (mode & MODE_APPLY) && pending > 0 && status == 0) {
        const size_t lb = (size_t)g * bs;
        for (int i = 0; i < pending; ++i) {
            const int node = E.leaf_node[lb + i];
            const int len = E.leaf_len[lb + i];
            const int ev = E.leaf_eval[lb + i];
            const int lnk = E.leaf_link[lb + i];
            const int cells = E.leaf_cells[lb + i];
            uint64_t lm[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) lm[s] = E.leaf_mask[(lb + i) * 4 + s];
            const int32_t *pth = E.path + (lb + i) * (size_t)pstride;
            for (int d = lane; d < len; d += 64) L.path[d] = pth[d];
            lds_sync();
            const bool terminal = ev < 0;
            float v = -1.0f;                                   // mcts.py:194-195
            if (!terminal) v = E.ev_value[ev];
            if (!value_in_fast_range(v)) slow_div = true;
            if (!expand(node, len, cells, lnk, terminal, lm, 0,
                        terminal ? nullptr : E.ev_prior + (size_t)ev * AZX_CELL_STRIDE, 0.0f))
                break;
            if (!pending_root) {
                path_apply(L.path, len, cells & 0xffff, -1, 1.0f, v, true, true);   // mcts.py:247-255
                if (lane == i) m_val = v;
            }
            lds_sync();
        }
        if (!pending_root && status == 0) search_value += np_sum_vals(pending);   // mcts.py:287
        pending = 0;
        pending_root = 0;
    }

    // =================================== SELECT (+ inline evaluate/expand/backup) =========
    const bool do_select = (mode & (MODE_SELECT | MODE_INLINE)) != 0;
    T_DECL
    while (do_select && batches_left > 0 && status == 0 && pending == 0) {
        if (root_link < 0) break;   // unevaluated or terminal root: nothing to search
        // The SIMD's issue arbiter favours its oldest waves, which then finish early and leave the
        // others to run alone with nothing to overlap their stalls.  Waves drop their own priority
        // as they progress, so the four games of a SIMD reach the end of the launch closer together
        // (measured: -9 % launch time; a finer, per-select dither of the four levels was slower).
#ifndef AZX_NO_PRIO
        if (num_batches > 0) {
#ifdef AZX_PRIO8   // diagnostic: eight progress levels dithered onto the four priorities
            const int q8 = (8 * batches_left - 1) / num_batches;  // 7 .. 0
            const int q = (q8 >> 1) + ((q8 & 1) & (batches_left & 1));
#else
            const int q = (4 * batches_left - 1) / num_batches;   // 3 .. 0
#endif
            if (q >= 3) __builtin_amdgcn_s_setprio(3);
            else if (q == 2) __builtin_amdgcn_s_setprio(2);
            else if (q == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        // ---- select_batch: bs sequential descents with virtual loss (mcts.py:62-70) ----
        // score_actions (mcts.py:119-136) op by op in float32 over RS register slots + np.argmax
        // (highest score, lowest index on ties, mcts.py:112): returns the winning lane-slot index.
        // Branch-free: every lane computes, lanes outside `mm` are masked out of the maximum.
        auto score_argmax = [&](auto rs_tag, const float4 *st, const float *Pn, const uint64_t *mm,
                                float sq) __attribute__((always_inline)) -> int {
            constexpr int RS = decltype(rs_tag)::value;
            uint32_t key[SLOTS], lkey = 0u;
            auto score_slots = [&](auto exact_tag) {
                constexpr bool kIeee = decltype(exact_tag)::value;
#pragma unroll
                for (int s = 0; s < RS; s += 2) {
                    const int s1 = s + 1 < RS ? s + 1 : s;
                    const f2 nvj = {st[s].x, st[s1].x};
                    const f2 tvj = {st[s].y, st[s1].y};
                    const f2 P = {Pn[s], Pn[s1]};
                    const f2 sq2 = {sq, sq}, one = {1.0f, 1.0f}, zero = {0.0f, 0.0f}, c2 = {c32, c32};
                    const f2 dg = one + nvj;
                    const f2 gap = kIeee ? div2_ieee(sq2, dg) : div2_unscaled(sq2, dg);   // mcts.py:132
                    const f2 U = (c2 * P) * gap;                                       // mcts.py:133
                    f2 dq;                                                             // clip(min=1)
                    dq.x = __builtin_amdgcn_fmed3f(nvj.x, 1.0f, 3.0e38f);
                    dq.y = __builtin_amdgcn_fmed3f(nvj.y, 1.0f, 3.0e38f);
                    const f2 W = -tvj;                                                 // search_tree.py:203
                    const f2 Q = kIeee ? div2_ieee(W, dq) : div2_unscaled(W, dq);     // mcts.py:134
                    const f2 score = (Q + U) + zero;                    // mcts.py:135; -0.0 -> +0.0
                    // order-preserving map float -> u32 (np.argmax compares values; equal
                    // floats <=> equal keys once -0.0 is folded into +0.0)
                    key[s] = lane_bit(mm[s]) ? f32_key(score.x) : 0u;
                    lkey = key[s] > lkey ? key[s] : lkey;
                    if (s1 != s) {
                        key[s1] = lane_bit(mm[s1]) ? f32_key(score.y) : 0u;
                        lkey = key[s1] > lkey ? key[s1] : lkey;
                    }
                }
            };
            if (slow_div) score_slots(std::true_type{});
            else score_slots(std::false_type{});
            int best_idx = 0x7fffffff;
            const uint32_t wmax = wave_max_u32(lkey);
#pragma unroll
            for (int s = RS - 1; s >= 0; --s) {
                const uint64_t eq = __ballot(key[s] == wmax) & mm[s];
                if (eq) best_idx = s * 64 + (int)__ffsll((long long)eq) - 1;
            }
            return best_idx;
        };
        auto sqrt_of = [&](float sq_tab, int sumn) __attribute__((always_inline)) -> float {
            float sq = late_vgpr(sq_tab);
            if (sumn >= AZX_SQRT_TAB) {                // beyond the table (rare): a real branch
                asm volatile("");                      // (the compiler would otherwise run the
                sq = sqrtf((float)sumn);               // 18-instruction sqrtf every time and select)
            }
            return sq;
        };

        // The root level of descent `sel` (children in registers by rank, Dirichlet noise, mcts.py:114):
        // pure function of the root state -- it is run for descent i + 1 while descent i's first child
        // block is on its way from HBM (its only input from descent i, the virtual loss on i's root
        // child, is applied as soon as that child is known).
        int pk_rank = 0, pk_cell = 0, pk_link = 0;
        float pk_nv = 0.0f;
        auto root_pick = [&](int sel, auto rs_tag) __attribute__((always_inline)) {
            constexpr int RS = decltype(rs_tag)::value;
            // sqrt(sum of the children's visits), mcts.py:132: the sum is the integer root_sumn.  A scalar
            // table load, requested first and consumed after the noise (a vector-register window of the
            // table made the compiler wait for ALL vector memory here, i.e. for the child block in
            // flight that this function is meant to overlap)
            const int sumn = root_sumn;
            const float sq_tab = c_sqrt[sumn < AZX_SQRT_TAB ? sumn : AZX_SQRT_TAB - 1];
            float nz[SLOTS] = {};
            const bool noisy = E.noise_scale != 0.0;
            float Pn[SLOTS];
#pragma unroll
            for (int s = 0; s < RS; ++s) Pn[s] = rst[s].z;
            if (noisy) {
                if (FAST || E.device_noise) {
                    // one stream word per (game, move, select): a Weyl sequence from the per-move
                    // base through one mixer (splitmix construction)
                    const uint32_t noise_stream = mix32(noise_base + (uint32_t)sel * 0x9E3779B9u);
                    dirichlet_noise<RS>(rootrm, lane, noise_stream, gconst, (float)E.noise_scale, nz);
                    // (1 - eps) * P, mcts.py:130: the same float32 product every select, so it
                    // is formed once per launch (rkp) and only the noise is added here
#pragma unroll
                    for (int s = 0; s < RS; ++s) Pn[s] = rkp[s] + nz[s];
                } else {
                    const double *row = E.noise + ((size_t)g * E.n_select + sel) * E.noise_stride;
#pragma unroll
                    for (int s = 0; s < RS; ++s)
                        if (lane_bit(rootrm[s]))
                            Pn[s] = (float)((double)(keep32 * Pn[s]) + E.noise_scale * row[64 * s + lane]);
                }
            }
            const float sq = sqrt_of(sq_tab, sumn);
            const int best_idx = score_argmax(rs_tag, rst, Pn, rootrm, sq);
            const int bl = best_idx & 63, bsl = best_idx >> 6;
#pragma unroll
            for (int s = 0; s < RS; ++s) {
                const int c_ = __builtin_amdgcn_readlane(rcell[s], bl);
                const int l_ = __builtin_amdgcn_readlane(__float_as_int(rst[s].w), bl);
                const float n_ = readlane_f(rst[s].x, bl);
                if (s == bsl) { pk_cell = c_; pk_link = l_; pk_nv = n_; }
            }
            pk_rank = best_idx;
            T_MARK(0)
            c_depth += 1;
            c_kint += (uint32_t)k_root;
        };
        auto root_pick_any = [&](int sel) __attribute__((always_inline)) {
            if (SLOTS >= 3 && k_root > 128) root_pick(sel, std::integral_constant<int, SLOTS>{});
            else if (k_root > 64) root_pick(sel, std::integral_constant<int, (SLOTS < 2 ? SLOTS : 2)>{});
            else root_pick(sel, std::integral_constant<int, 1>{});
        };

        root_pick_any(select_count);
        memo_ok = true;
        for (int i = 0; i < bs; ++i) {
            // snapshot/restore (search_tree.py:150-154): the descent only tracks which cells are
            // occupied (`all`), scalar bit-ors; the win test runs once, at a new leaf (interior
            // nodes are not won)
            uint64_t all[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) all[s] = rootall[s];
            // ---- the root step of this descent, chosen by root_pick (search_tree.py:306-308) ----
            const int cell0 = pk_rank;                        // (the root child's rank: rst is by rank)
            int cellL = pk_cell;
            int link = root_link, node = root_link + pk_rank, child_link = pk_link, depth = 1;
            int mover = 3 - root.color;                       // colour placing the next stone
            float cur_nv = 0.0f;               // num_visits (incl. virtual) of the node being scored
            float cnv = pk_nv, ctv = 0.f;
            float sq_next_pend = 0.0f;
            int slot1 = -1;                                    // cache entry of the depth-2 node, once known
            {
                const uint64_t bit = 1ull << (pk_cell & 63);
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
                    if (s == (pk_cell >> 6)) all[s] |= bit;
            }
            if (lane == 0) L.path[i * pstride] = node | (pk_cell << 24);
            // its virtual loss (mcts.py:68) goes on at once: the next root level needs nothing else
            root_child_add(cell0, 1.0f, 1.0f);
            T_MARK(2)

            // ---- levels below the root: lane = board cell, children from HBM ---------------------
            float4 pst[SLOTS];                  // the requested child block
            int prk[SLOTS];
            bool pshort = false;
            auto masks_of_all = [&]() __attribute__((always_inline)) -> Masks<SLOTS> {
                Masks<SLOTS> mk;
                int k = 0;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    mk.m[s] = ~all[s];
                    mk.base[s] = k;
                    k += popc64(mk.m[s]);
                }
                mk.k = k;
                return mk;
            };
            auto deeper_issue = [&]() __attribute__((always_inline)) {
                link = child_link;
                cur_nv = cnv;
                // sqrt(visits - 1) of this node: requested here, consumed after the children arrive
                const int sn = (int)cur_nv - 1;
                sq_next_pend = c_sqrt[sn < 0 ? 0 : (sn < AZX_SQRT_TAB ? sn : AZX_SQRT_TAB - 1)];
                // A node visited once (its own expansion) has only unvisited children: no load
                pshort = cur_nv == 1.0f;
                if (!pshort) {
                    const Masks<SLOTS> mk = masks_of_all();
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) prk[s] = mk.base[s] + rank_below(mk.m[s]);
                    // every lane loads -- occupied cells re-read child 0 and are masked out of the
                    // argmax below -- so there is no divergent region around the loads and the
                    // slots' requests are in flight together (one HBM round trip per level)
                    const float4 *cblk = reinterpret_cast<const float4 *>(arena + link);
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) pst[s] = cblk[lane_bit(mk.m[s]) ? prk[s] : 0];
                }
            };
            auto deeper_finish = [&]() __attribute__((always_inline)) {
                const Masks<SLOTS> mk = masks_of_all();
                int best_cell = 0x7fffffff, child_rank = 0;
                cnv = 0.f;
                ctv = 0.f;
                if (pshort) {
                    // (sum of the children's visits == its visits - 1 == 0): every score is -0 + 0, so
                    // np.argmax takes child 0, an unevaluated leaf.  No scoring.
#pragma unroll
                    for (int s = SLOTS - 1; s >= 0; --s)
                        if (mk.m[s]) best_cell = s * 64 + (int)__ffsll((long long)mk.m[s]) - 1;
                    child_link = AZX_LINK_UNEVAL;
                } else {
                    // sqrt(sum of the children's visits), mcts.py:132: every evaluated node has been
                    // visited once more than its children together (its own expansion), virtual
                    // losses included since they mark node and child alike.
                    const int sumn = (int)cur_nv - 1;
                    // newer (num_visits, total_value) of cached children override HBM
                    uint64_t pm = __ballot(lane < n_c && c_id >= link && c_id < link + mk.k);
                    while (pm) {
                        const int j = (int)__ffsll((long long)pm) - 1;
                        pm &= pm - 1;
                        const int r = __builtin_amdgcn_readlane(c_id, j) - link;
                        const float pnv = readlane_f(c_nv, j), ptv = readlane_f(c_tv, j);
#pragma unroll
                        for (int s = 0; s < SLOTS; ++s)
                            if (lane_bit(mk.m[s]) && prk[s] == r) { pst[s].x = pnv; pst[s].y = ptv; }
                    }
                    float Pn[SLOTS];
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) Pn[s] = pst[s].z;
                    const float sq = sqrt_of(sq_next_pend, sumn);
                    const int best_idx = score_argmax(std::integral_constant<int, SLOTS>{}, pst, Pn, mk.m, sq);
                    const int bl = best_idx & 63, bsl = best_idx >> 6;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        const int r_ = __builtin_amdgcn_readlane(prk[s], bl);
                        const int l_ = __builtin_amdgcn_readlane(__float_as_int(pst[s].w), bl);
                        const float n_ = readlane_f(pst[s].x, bl), t_ = readlane_f(pst[s].y, bl);
                        if (s == bsl) { child_rank = r_; child_link = l_; cnv = n_; ctv = t_; }
                    }
                    best_cell = best_idx;
                }
                T_MARK(1)
                c_depth += 1;
                c_kint += (uint32_t)mk.k;
                node = link + child_rank;
                {                                              // deeper path nodes enter the cache
                    const int ce = cache_find(node, true, cnv, ctv);
                    if (depth == 1) slot1 = ce;
                }
                cellL = best_cell;
                if (lane == 0) L.path[i * pstride + depth] = node | (best_cell << 24);
                {                                              // search_tree.py:306-308, stones only
                    const uint64_t bit = 1ull << (best_cell & 63);
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        if (s == (best_cell >> 6)) all[s] |= bit;
                    mover = 3 - mover;
                }
                T_MARK(2)
                depth += 1;
            };
            const bool go_deeper = child_link >= 0;            // < 0: a leaf, unevaluated or terminal
            if (go_deeper) deeper_issue();
            __builtin_amdgcn_sched_barrier(0);                 // the requests leave before the next root level
            if (i + 1 < bs) root_pick_any(select_count + 1);   // in the shadow of that load
            // (Round 5: the NEXT descent's first child block is known here, a whole descent before its load -- touching
            // one dword per 128-byte line of it now, so that the load finds the lines in L2, was built and measured:
            // 1.327 vs 1.217 ms per move, three alternating runs on one box, profiles/r5_tree_touch_ab.txt.  The stamps
            // say why: waiting for and scoring child blocks is 17.6 % of a wave's cycles, and the touch's own address
            // arithmetic + one more vector-memory instruction per descent cost more than the latency they hide.)
            if (go_deeper) {
                if (!pshort) {
                    // the child block is first touched HERE (the compiler otherwise copies parts of it
                    // into other registers right behind the loads and waits for them there)
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        asm volatile("" : "+v"(pst[s].x), "+v"(pst[s].y), "+v"(pst[s].z), "+v"(pst[s].w));
                }
                deeper_finish();
                while (child_link >= 0) {
                    deeper_issue();
                    deeper_finish();
                }
            }
            select_count += 1;
            c_selects += 1;
            // ---- the leaf position: winner (hex.py:204-231) and legal moves ---------------------
            const int last = 3 - mover;                       // colour of the stone just placed
            int winner = last;                                // a terminal link: the last mover won
            if (child_link == AZX_LINK_UNEVAL) {
                if (depth <= 2) {
                    // `last` has one stone on the path: the root's winning-move mask of that colour
                    uint64_t wmask = 0ull;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        if (s == (cellL >> 6)) wmask = last == 1 ? winm[0][s] : winm[1][s];
                    winner = ((wmask >> (cellL & 63)) & 1ull) ? last : 0;
                } else {
                    // several stones of `last`: replay them on a copy of the root groups
                    lds_sync();
                    HexWave<SLOTS> tmp = root;
                    for (int d = (depth - 1) & 1; d < depth; d += 2) {
                        const int cell = (L.path[i * pstride + d] >> 24) & 0xff;
                        tmp.color = last;
                        tmp.step(cell, N, lane);
                    }
                    winner = tmp.winner;
                }
            }
            uint64_t lmask[SLOTS];
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) lmask[s] = ~all[s];
            if (need_colors) {
                // the leaf board: root colours plus the path's stones, alternating from the root mover
                lds_sync();
                uint32_t v[SLOTS];
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) v[s] = root.c[s] & 3u;
                for (int d = 0; d < depth; ++d) {
                    const int cell = (L.path[i * pstride + d] >> 24) & 0xff;
                    const uint32_t colour = (uint32_t)((d & 1) ? 3 - root.color : root.color);
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        if (s * 64 + lane == cell) v[s] = colour;
                }
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int cell = s * 64 + lane;
                    if (cell < ncells) L.colors[i * AZX_CELL_STRIDE + cell] = (unsigned char)v[s];
                }
            }
            if (lane == i) {
                m_node = node;
                m_len = depth;
                m_link = child_link;
                m_tm = (winner != 0 ? 1 : 0) | (mover << 1);
                m_cells = cell0 | (cellL << 16);
                m_slot1 = slot1;
                if (inline_eval) {                             // one prior for all children: only their number
                    int kl = 0;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) kl += popc64(lmask[s]);
                    m_k = kl;
                } else {
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) m_mask[s] = lmask[s];
                }
            }
            lds_sync();
            // ... and apply the virtual loss to its path (mcts.py:68, :79-92)
            // (the root child's share went on right after the root step)
            path_apply(L.path + i * pstride, depth, -1, slot1, 1.0f, 1.0f, false, false);
            T_MARK(3)
        }
        int nu = 0;
        if (FAST) {
            // With the uniform evaluator every backed-up value is 0 or -1, so every num_visits and
            // total_value is a small integer and float32 addition on them is exact in any order: the
            // undo of the virtual losses (mcts.py:72), deduplicate_leaves (:139-152), expand_batch
            // (:226-239) and backup_batch (:242-255) are folded into ONE pass over the batch's paths
            // with the net change per node -- a first-seen leaf keeps the visit its virtual loss
            // added and gives back the +1 of total_value (plus the alternating -1 of a terminal
            // leaf), a duplicate gives back both -- instead of two passes of separate updates.
            // Node ids are still handed out in list order of the first occurrences.
            for (int i = 0; i < bs; ++i) {
                const int id = rl(m_node, i);
                const bool dup = __ballot(lane < i && m_node == id) != 0ull;
                const int len = rl(m_len, i), cells = rl(m_cells, i);
                bool term = false;
                if (!dup) {
                    term = (rl(m_tm, i) & 1) != 0;
                    const int k = rl(m_k, i);
                    if (term) c_term += 1; else c_evals += 1;
                    if (!expand(id, len, cells, rl(m_link, i), term, nullptr, k, nullptr,
                                (!term && k) ? inline_prior(k, len) : 0.0f))
                        break;
                    if (lane == nu) m_val = term ? -1.0f : 0.0f;
                    nu += 1;
                    root_nv += 1.0f;
                    if (term) root_tv += (len & 1) ? 1.0f : -1.0f;
                }
                T_MARK(4)
                const float dvn = dup ? -1.0f : 0.0f;
                // total_value change of the path node at depth d + 1
                auto tvd = [&](int d) -> float {
                    return term ? (((len - 1 - d) & 1) ? 0.0f : -2.0f) : -1.0f;
                };
                root_child_add(cells & 0xffff, dvn, tvd(0));
                const int32_t *pth = L.path + i * pstride;
                const int s1 = rl(m_slot1, i);
                for (int d = 1; d < len; ++d) {
                    const int slot = (d == 1 && s1 >= 0 && memo_ok) ? s1
                                                                    : cache_find(pth[d] & 0xffffff, false, 0.f, 0.f);
                    if (lane == slot) { c_nv += dvn; c_tv += tvd(d); }
                }
                T_MARK(5)
            }
            batches_left -= 1;
            wave_mem_sync();   // the children written by rank above are read by other lanes from here on
            if (status == 0) search_value += np_sum_vals(nu);   // mcts.py:287
            T_MARK(5)
            continue;
        }
        // undo the virtual losses in list order (mcts.py:72)
        for (int i = 0; i < bs; ++i)
            path_apply(L.path + i * pstride, rl(m_len, i), rl(m_cells, i) & 0xffff, rl(m_slot1, i), -1.0f, -1.0f,
                       false, false);
        // deduplicate_leaves: keep first occurrence by node id (mcts.py:139-152)
        for (int i = 0; i < bs; ++i) {
            const int id = rl(m_node, i);
            const bool dup = __ballot(lane < i && m_node == id) != 0ull;
            if (!dup) {
                if (lane == nu) m_uidx = i;
                nu += 1;
            }
        }
        batches_left -= 1;
        T_MARK(4)

        if (inline_eval) {
            // evaluate_batch + expand_batch + backup_batch for the inline (uniform) evaluator
            for (int u = 0; u < nu; ++u) {
                const int i = rl(m_uidx, u);
                const int tm = rl(m_tm, i);
                const bool terminal = (tm & 1) != 0;
                const int k = rl(m_k, i);
                float v = -1.0f;
                if (!terminal) {
                    v = (!FAST && E.evaluator == AZX_EVAL_UNIFORM_HASH)
                            ? hash_value(L.colors + i * AZX_CELL_STRIDE, tm >> 1) : 0.0f;
                    c_evals += 1;
                    if (!value_in_fast_range(v)) slow_div = true;
                } else {
                    c_term += 1;
                }
                const int lf_len = rl(m_len, i), lf_cells = rl(m_cells, i);
                if (!expand(rl(m_node, i), lf_len, lf_cells, rl(m_link, i), terminal, nullptr, k, nullptr,
                            (!terminal && k) ? inline_prior(k, lf_len) : 0.0f))
                    break;
                path_apply(L.path + i * pstride, lf_len, lf_cells & 0xffff, rl(m_slot1, i), 1.0f, v, true, true);
                if (lane == u) m_val = v;
            }
            wave_mem_sync();   // the children written by rank above are read by other lanes from here on
            if (status == 0) search_value += np_sum_vals(nu);   // mcts.py:287
            T_MARK(5)
        } else {
            // hand the unique leaves to the evaluator: scratch + packed requests
            int n_nt = 0;
            for (int u = 0; u < nu; ++u) n_nt += (rl(m_tm, rl(m_uidx, u)) & 1) ? 0 : 1;
            int e0 = 0;
            if (lane == 0 && n_nt) e0 = atomicAdd(E.n_eval, n_nt);
            e0 = __builtin_amdgcn_readfirstlane(e0);
            const size_t lb = (size_t)g * bs;
            int e = e0;
            for (int u = 0; u < nu; ++u) {
                const int i = rl(m_uidx, u);
                const int tm = rl(m_tm, i);
                const bool terminal = (tm & 1) != 0;
                const int len = rl(m_len, i);
                int32_t *pth = E.path + (lb + u) * (size_t)pstride;
                for (int d = lane; d < len; d += 64) pth[d] = L.path[i * pstride + d];
                if (lane == 0) {
                    E.leaf_node[lb + u] = rl(m_node, i);
                    E.leaf_len[lb + u] = len;
                    E.leaf_link[lb + u] = rl(m_link, i);
                    E.leaf_cells[lb + u] = rl(m_cells, i);
                    E.leaf_eval[lb + u] = terminal ? -1 : e;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) E.leaf_mask[(lb + u) * 4 + s] = rl64(m_mask[s], i);
                }
                if (!terminal) {
                    emit_request(e, g * bs + u, L.colors + i * AZX_CELL_STRIDE, tm >> 1);
                    e += 1;
                    c_evals += 1;
                } else {
                    c_term += 1;
                }
            }
            pending = nu;
        }
    }

    // =================================== epilogue: write the on-chip state back ===========
    T_FLUSH
    cache_flush();
    if (root_link >= 0) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (rempty[s]) *reinterpret_cast<float4 *>(arena + root_link + s * 64 + lane) = rst[s];
    }
    if (lane == 0) {
        *reinterpret_cast<float2 *>(arena + root_id) = make_float2(root_nv, root_tv);
        th->num_nodes = num_nodes;
        th->status = status;
        th->batches_left = batches_left;
        th->pending = pending;
        th->pending_root = pending_root;
        th->select_count = select_count;
        th->search_value = search_value;
        if (!FAST) th->slow_div = slow_div ? 1 : 0;
    }
    // the six per-game tallies, one lane each (as six scalar read-modify-writes the compiler waited
    // for each load in turn: six memory round trips at the end of every launch)
    static_assert(CTR_SELECTS == 0 && CTR_SUM_DEPTH == 1 && CTR_SUM_K_INT == 2 && CTR_SUM_K_LEAF == 3 &&
                  CTR_EVALS == 4 && CTR_TERM_EVALS == 5, "tally order");
    if (lane < 6) {
        const uint32_t add = lane == 0 ? c_selects : lane == 1 ? c_depth : lane == 2 ? c_kint
                           : lane == 3 ? c_kleaf : lane == 4 ? c_evals : c_term;
        if (add) E.counters[(size_t)g * CTR_COUNT + lane] += add;
    }
}

template <int SLOTS, bool FAST>
__global__ __launch_bounds__(64) AZX_MCTS_ATTR void k_mcts(DevEngine E, int mode_arg, int num_batches) {
    mcts_body<SLOTS, FAST>(E, mode_arg, num_batches);
}

// ============================================================================================
// reset + replay: HexGame.reset / step (hex.py:47-49, :172-179), SearchTree.reset
// (search_tree.py:59-71).  One wavefront per listed slot.
// ============================================================================================
template <int SLOTS>
__device__ __forceinline__ void tree_reset(const DevEngine &E, int g, TreeHdr *th, int k, int lane) {
    Node *arena = E.arena[th->arena] + (size_t)g * E.cap;
    if (lane == 0) {
        Node nd;
        nd.nv = 0.0f; nd.tv = 0.0f; nd.pp = 1.0f; nd.link = AZX_LINK_UNEVAL;
        arena[0] = nd;
        th->num_nodes = 1;
        th->root_id = 0;
        th->root_k = k;
        th->k0 = k;
        th->status = 0;
        th->batches_left = 0;
        th->pending = 0;
        th->pending_root = 0;
        th->select_count = 0;
        th->search_value = 0.0f;
        th->slow_div = 0;
        th->defer_compact = 0;
        th->dropped = 0;
    }
}

template <int SLOTS>
__global__ __launch_bounds__(64) void k_reset(DevEngine E, const int32_t *slots, int n_slots,
                                              const int32_t *moves, const int32_t *n_moves,
                                              int stride, int assign_uid) {
    const int lane = threadIdx.x;
    const int idx = blockIdx.x;
    if (idx >= n_slots) return;
    const int g = slots ? slots[idx] : idx;
    HexWave<SLOTS> h;
    h.clear();
    h.geom(E.N, lane);
    int ply = 0;
    if (moves) {
        const int nm = n_moves[idx];
        for (int p = 0; p < nm; ++p) {
            const int mv = moves[(size_t)idx * stride + p];
            h.step(mv - 1, E.N, lane);
            ply += 1;
        }
    }
    h.store(E.cells + (size_t)g * SLOTS * 64, lane);
    const Masks<SLOTS> mk = make_masks<SLOTS>(h, lane, E.ncells);
    GameHdr *gh = E.ghdr + g;
    TreeHdr *th = E.thdr + g;
    if (lane == 0) {
        gh->color = h.color;
        gh->winner = h.winner;
        gh->ply = ply;
        gh->active = 1;
        gh->move_id = -1;
        gh->n_rows = 0;
        gh->parked = 0;
        gh->ply0 = ply;
        if (assign_uid) { gh->uid = game_uid(E, g, gh->gen); gh->gen += 1; }
        th->arena = 0;
    }
    __builtin_amdgcn_s_waitcnt(0);
    tree_reset<SLOTS>(E, g, th, mk.k, lane);
}

// ============================================================================================
// advance: HexGame.step + SearchTree.move (hex.py:172-179, search_tree.py:115-132), with the
// kept subtree compacted into the other arena (the reference never reclaims nodes).
// In play mode a finished game is appended to the output queue and the slot restarts.
// ============================================================================================
template <int SLOTS>
__device__ __forceinline__ void advance_body(const DevEngine &E, const int32_t *move_ids, int play_mode) {
    __shared__ int sh_old[64];         // compaction: old first-child ids of a group's parents, by rank
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    GameHdr *gh = E.ghdr + g;
    TreeHdr *th = E.thdr + g;
    // play_mode 2 = unpark: only the slots whose finished game found the harvest queue full take part,
    // and they go straight to the harvest (the move was stepped when the game finished)
    const bool parked = gh->parked != 0;
    if (play_mode == 2 ? !parked : !gh->active) return;
    int status = th->status;
    const int mid = move_ids ? move_ids[g] : gh->move_id;
    if (!parked && mid < 0 && status == 0) return;

    HexWave<SLOTS> h;
    h.load(E.cells + (size_t)g * SLOTS * 64, lane);
    h.geom(E.N, lane);
    h.color = gh->color;
    h.winner = gh->winner;
    int ply = gh->ply;
    bool finished = parked, errored = !parked && status != 0;

    if (!errored && !parked) {
        const Masks<SLOTS> mk = make_masks<SLOTS>(h, lane, E.ncells);
        // the mid-th legal move in ascending tile order (search_tree.py:306)
        int cell = -1;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const bool empty = lane_bit(mk.m[s]);
            const int rk = mk.base[s] + rank_below(mk.m[s]);
            const uint64_t hit = __ballot(empty && rk == mid);
            if (hit) cell = s * 64 + (int)__ffsll((long long)hit) - 1;
        }
        if (cell < 0) return;   // illegal move_id: leave the slot untouched
        h.step(cell, E.N, lane);
        ply += 1;
        h.store(E.cells + (size_t)g * SLOTS * 64, lane);
        const Masks<SLOTS> nmk = make_masks<SLOTS>(h, lane, E.ncells);
        finished = h.winner != 0;

        // ---- SearchTree.move (search_tree.py:115-132) ----
        Node *src = E.arena[th->arena] + (size_t)g * E.cap;
        const int root_id = th->root_id;
        const int root_link = src[root_id].link;
        int child = -1, clink = AZX_LINK_UNEVAL;
        if (root_link >= 0) {
            child = root_link + mid;
            clink = src[child].link;
        }
        if (child < 0 || clink == AZX_LINK_UNEVAL) {
            tree_reset<SLOTS>(E, g, th, nmk.k, lane);           // step to the unknown
        } else if ((E.flags & AZX_FLAG_NO_COMPACT) ||
                   (long long)th->num_nodes + (long long)(E.selects_per_search + 1) * nmk.k <= (long long)E.cap) {
            // re-root in place, like the reference (search_tree.py:127-130).  The arena is only
            // compacted when the next search could run out of nodes (each select_leaf call and the
            // root evaluation expand at most one node with at most k children).
            if (lane == 0) { th->root_id = child; th->root_k = nmk.k; }
        } else {
            if (play_mode) {
                // play mode: re-root in place now and leave the copy to the next search launch, where
                // it runs beside the other games' searches instead of holding up this whole launch
                // (every launch has some games compacting; it was 70 % of k_advance's time)
                if (lane == 0) { th->root_id = child; th->root_k = nmk.k; th->defer_compact = 1; }
            } else {
                compact_tree(E, g, th, child, nmk.k, lane, sh_old);
            }
        }
        if (lane == 0) {
            gh->color = h.color;
            gh->winner = h.winner;
            gh->ply = ply;
            gh->move_id = -1;
            E.counters[(size_t)g * CTR_COUNT + CTR_PLIES] += 1ull;
        }
    }

    if (!play_mode || !(finished || errored)) return;

    // ---- play mode: harvest the finished game (play_game.py:59-67) and restart the slot ----
    const int rows = gh->n_rows;
    bool restart = true;
    if (finished) {
        const int result = h.winner == 2 ? 1 : 3;              // hex.py:161-170
        unsigned long long pos = 0;
        if (lane == 0) pos = atomicAdd(E.q_count, (unsigned long long)rows);
        pos = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(pos >> 32)) << 32) |
              (unsigned int)__builtin_amdgcn_readfirstlane((int)pos);
        if (!E.q_ring && pos + rows > (unsigned long long)E.q_cap) {
            // queue full: give the reservation back and park the slot until the host drains
            if (lane == 0) {
                atomicAdd(E.q_count, (unsigned long long)(-(long long)rows));
                gh->active = 0;
                gh->parked = 1;
            }
            restart = false;
        } else {
            // the game's rows are contiguous at the source, and at the destination unless the ring
            // wraps inside them: bulk 16-byte copies, sixteen per lane in flight
            const size_t q0 = (size_t)(pos % (unsigned long long)E.q_cap);
            const int first = (int)min((unsigned long long)rows, (unsigned long long)E.q_cap - q0);   // rows before the wrap
            auto copy16 = [&](void *dst, const void *src, int n16) {
                uint4 *d = reinterpret_cast<uint4 *>(dst);
                const uint4 *sp = reinterpret_cast<const uint4 *>(src);
                // sixteen 16-byte pieces in flight per lane: the launch lasts as long as its longest
                // finished game takes to copy (one wave, ~100 KB), i.e. as many memory round trips
                constexpr int U = 16;
                for (int i0 = 0; i0 < n16; i0 += 64 * U) {
                    uint4 v[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = min(i0 + u * 64 + lane, n16 - 1);   // (the tail repeats the last piece)
                        v[u] = sp[i];
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = i0 + u * 64 + lane;
                        if (i < n16) d[i] = v[u];
                    }
                }
            };
            const size_t sr0 = (size_t)g * E.ncells * AZX_CELL_STRIDE;
            for (int part = 0; part < 2; ++part) {
                const int r0 = part ? first : 0, nr = part ? rows - first : first;
                if (nr <= 0) continue;
                const size_t qd = part ? 0 : q0;
                copy16(E.q_board + qd * AZX_CELL_STRIDE, E.row_board + sr0 + (size_t)r0 * AZX_CELL_STRIDE,
                       nr * (AZX_CELL_STRIDE / 16));
                copy16(E.q_prob + qd * AZX_CELL_STRIDE, E.row_prob + sr0 + (size_t)r0 * AZX_CELL_STRIDE,
                       nr * (AZX_CELL_STRIDE / 4));
            }
            const int64_t uid = gh->uid;
            const int ply0 = gh->ply0;                         // row r was recorded at ply ply0 + r
            for (int r = lane; r < rows; r += 64) {
                const size_t q = (size_t)((pos + r) % (unsigned long long)E.q_cap);
                E.q_color[q] = (ply0 + r) & 1;
                E.q_k[q] = E.row_k[(size_t)g * E.ncells + r];
                float rew = (float)(result - 2);               // play_game.py:64-65
                if ((ply0 + r) & 1) rew = -rew;
                E.q_reward[q] = rew;
                E.q_uid[q] = uid;
                const float4 *ms = reinterpret_cast<const float4 *>(E.row_meta) + ((size_t)g * E.ncells + r) * 2;
                float4 mt = ms[0];
                mt.w = r == 0 ? 1.0f : 0.0f;                   // marks the first row of a game
                reinterpret_cast<float4 *>(E.q_meta)[q * 2] = mt;
                reinterpret_cast<float4 *>(E.q_meta)[q * 2 + 1] = ms[1];
            }
            if (lane == 0) {
                E.counters[(size_t)g * CTR_COUNT + CTR_GAMES] += 1ull;
                E.counters[(size_t)g * CTR_COUNT + CTR_ROWS] += (unsigned long long)rows;
                float last = (float)(result - 2);
                if ((ply0 + rows - 1) & 1) last = -last;
                E.stat_sums[(size_t)g * 8 + 3] += (double)last;       // metrics['reward']
                E.stat_sums[(size_t)g * 8 + 4] += (double)ply;        // game length from the empty board
            }
        }
    } else {
        if (lane == 0) E.counters[(size_t)g * CTR_COUNT + CTR_ERRORS] += 1ull;   // parallel_player.py:73-76
    }
    if (restart) {
        HexWave<SLOTS> z;
        z.clear();
        z.store(E.cells + (size_t)g * SLOTS * 64, lane);
        if (lane == 0) {
            gh->color = 1;
            gh->winner = 0;
            gh->ply = 0;
            gh->move_id = -1;
            gh->n_rows = 0;
            gh->ply0 = 0;
            if (parked) { gh->parked = 0; gh->active = 1; }
            gh->uid = game_uid(E, g, gh->gen);
            gh->gen += 1;
        }
        __builtin_amdgcn_s_waitcnt(0);
        tree_reset<SLOTS>(E, g, th, E.ncells, lane);
    }
}

// ============================================================================================
// root statistics: root.move_stats (search_tree.py:192-204) dense by child index
// ============================================================================================
template <int SLOTS>
__global__ __launch_bounds__(64) void k_advance(DevEngine E, const int32_t *move_ids, int play_mode) {
    advance_body<SLOTS>(E, move_ids, play_mode);
}

template <int SLOTS>
__global__ __launch_bounds__(64) void k_gather_root(DevEngine E, int32_t *k_out, int32_t *legal,
                                                    float *cv, float *cw, float *cp, float *rv,
                                                    float *rw, int32_t *nn, float *sv) {
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    GameHdr *gh = E.ghdr + g;
    TreeHdr *th = E.thdr + g;
    HexWave<SLOTS> h;
    h.load(E.cells + (size_t)g * SLOTS * 64, lane);
    h.color = gh->color;
    h.winner = gh->winner;
    const Masks<SLOTS> mk = make_masks<SLOTS>(h, lane, E.ncells);
    const Node *arena = E.arena[th->arena] + (size_t)g * E.cap;
    const Node rootn = arena[th->root_id];
    const size_t ob = (size_t)g * E.ncells;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        if (lane_bit(mk.m[s]) && mk.k > 0) {
            const int rk = mk.base[s] + rank_below(mk.m[s]);
            if (legal) legal[ob + rk] = s * 64 + lane + 1;
            if (rootn.link >= 0) {
                const Node c = arena[rootn.link + rk];
                if (cv) cv[ob + rk] = c.nv;
                if (cw) cw[ob + rk] = c.tv;
                if (cp) cp[ob + rk] = c.pp;
            }
        }
    }
    if (lane == 0) {
        if (k_out) k_out[g] = mk.k;
        if (rv) rv[g] = rootn.nv;
        if (rw) rw[g] = rootn.tv;
        if (nn) nn[g] = th->num_nodes;
        if (sv) sv[g] = th->search_value;
    }
}

// ============================================================================================
// throughput mode: Policy.choose_action's move draw (policy.py:142-160) + play_game's data
// collection (play_game.py:81-98) on the device.  as_distribution (search_tree.py:327-344):
// T>0: p ~ n^(1/T); T==0: uniform over the most-visited children.
// ============================================================================================
template <int SLOTS>
__device__ __forceinline__ void choose_body(const DevEngine &E) {
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    GameHdr *gh = E.ghdr + g;
    TreeHdr *th = E.thdr + g;
    if (!gh->active || th->status != 0) return;
    HexWave<SLOTS> h;
    h.load(E.cells + (size_t)g * SLOTS * 64, lane);
    h.color = gh->color;
    h.winner = gh->winner;
    const Masks<SLOTS> mk = make_masks<SLOTS>(h, lane, E.ncells);
    const Node *arena = E.arena[th->arena] + (size_t)g * E.cap;
    const Node rootn = arena[th->root_id];
    if (rootn.link < 0 || mk.k == 0) return;
    const int ply = gh->ply;
    const float T = (ply >= E.exploration_depth) ? 0.0f : E.temperature;   // policy.py:142-149

    float nv[SLOTS], w[SLOTS];
    int rk[SLOTS];
    float mx = 0.0f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        nv[s] = 0.0f;
        rk[s] = mk.base[s] + rank_below(mk.m[s]);
        if (lane_bit(mk.m[s])) nv[s] = arena[rootn.link + rk[s]].nv;
        mx = fmaxf(mx, nv[s]);
    }
    mx = wave_max(mx);
    float nv_sum = 0.0f;                                 // visit counts are integers: the sum is exact in any order
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) nv_sum += nv[s];
    nv_sum = wave_sum(nv_sum);
    float tot = 0.0f;
    int width = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const bool empty = lane_bit(mk.m[s]);
        float x = 0.0f;
        if (empty) {
            if (T > 0.0f) x = (nv[s] > 0.0f) ? (T == 1.0f ? nv[s] : __powf(nv[s], 1.0f / T)) : 0.0f;
            else x = (nv[s] == mx) ? 1.0f : 0.0f;
            width += nv[s] > 0.0f;
        }
        w[s] = x;
        tot += x;
    }
    tot = wave_sum(tot);
    width = wave_sum_i(width);
    // inclusive prefix sums in child order (slot-major, lane-minor)
    uint32_t r[4];
    game_rng(E, gh->uid).gen(0xC0FFEEu, (uint32_t)ply, 0u, 0x4D4F5645u, r);
    const float target = u01(r[0]) * tot;
    float run = 0.0f;
    int chosen = -1;
    float chosen_w = 0.0f;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        float inc = w[s];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        const float cum = run + inc;
        const bool cand = lane_bit(mk.m[s]) && w[s] > 0.0f && cum > target;
        const uint64_t cm = __ballot(cand);
        if (cm && chosen < 0) {
            const int l = (int)__ffsll((long long)cm) - 1;
            chosen = __builtin_amdgcn_readlane(rk[s], l);
            chosen_w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[s]), l));
        }
        run += __shfl(inc, 63, 64);
    }
    if (chosen < 0) {   // rounding left the target at the very top: take the last positive child
#pragma unroll
        for (int s = SLOTS - 1; s >= 0; --s) {
            const uint64_t cm = __ballot(lane_bit(mk.m[s]) && w[s] > 0.0f);
            if (cm && chosen < 0) {
                const int l = 63 - __clzll((long long)cm);
                chosen = __builtin_amdgcn_readlane(rk[s], l);
                chosen_w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[s]), l));
            }
        }
    }
    // replay row (play_game.py:92-94): pre-move board and moves_prob
    const int row = gh->n_rows;
    const size_t rb = ((size_t)g * E.ncells + row) * AZX_CELL_STRIDE;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int cell = s * 64 + lane;
        if (cell < AZX_CELL_STRIDE) E.row_board[rb + cell] = (unsigned char)(h.c[s] & 3u);
        if (lane_bit(mk.m[s])) E.row_prob[rb + rk[s]] = w[s] / tot;
    }
    for (int o = mk.k + lane; o < AZX_CELL_STRIDE; o += 64) E.row_prob[rb + o] = 0.0f;
    if (lane == 0) {
        E.row_k[(size_t)g * E.ncells + row] = mk.k;
        gh->n_rows = row + 1;
        gh->move_id = chosen;
        const float logp = __logf(chosen_w / tot);
        const float sval = th->search_value / (float)E.selects_per_search;   // mcts.py:291 (as azx_get_root reports it)
        E.stat_sums[(size_t)g * 8 + 0] += (double)sval;
        E.stat_sums[(size_t)g * 8 + 1] += (double)width;
        E.stat_sums[(size_t)g * 8 + 2] += (double)logp;
        // search_tree.py:109-112: width, mean child visits, nodes ever allocated, children
        float4 *meta = reinterpret_cast<float4 *>(E.row_meta) + ((size_t)g * E.ncells + row) * 2;
        meta[0] = make_float4(sval, (float)width, logp, 0.0f);
        meta[1] = make_float4(nv_sum / (float)mk.k, (float)(th->num_nodes + th->dropped), (float)mk.k, 0.0f);
    }
}

// ============================================================================================
// rules only: replay move lists, reporting per-ply result / legal count / empties mask
// ============================================================================================
template <int SLOTS>
__global__ __launch_bounds__(64) void k_choose(DevEngine E) {
    choose_body<SLOTS>(E);
}

// Throughput mode with the inline uniform evaluator: `steps` whole moves of every game in ONE launch.
// A wave owns its game from search to move draw to game step (and harvest / restart), and nothing
// but the harvest queue's counter is shared between games, so there is no reason to line all games
// up at three launch boundaries per move: each wave runs its own search -> choose -> advance loop,
// and the harvest copies and arena compactions of some games overlap the searches of the others.
template <int SLOTS>
__global__ __launch_bounds__(64) AZX_MCTS_ATTR void k_play(DevEngine E, int num_batches, int steps) {
#ifdef AZX_STAMP_PLAY     // diagnostic build: per-wave cycles in the search, the move draw and the game step (slots 10-12)
    unsigned long long tp_[3] = {0, 0, 0};
#define TP(i, stmt) { const unsigned long long a_ = __builtin_amdgcn_s_memtime(); stmt; tp_[i] += __builtin_amdgcn_s_memtime() - a_; }
#else
#define TP(i, stmt) stmt;
#endif
    for (int s = 0; s < steps; ++s) {
        TP(0, (mcts_body<SLOTS, true>(E, MODE_BEGIN | MODE_INLINE, num_batches, s == 0)))
        wave_mem_sync();
        // The move draw and the game step read their engine fields (queue, row and statistics pointers: ~50
        // scalar registers) from the kernel-argument segment again, through a pointer the compiler cannot see
        // through: loaded at kernel entry they stay live -- spilled to VGPR lanes and reloaded -- across the
        // whole search.
#ifdef AZX_PLAY_EARLY_ARGS     // diagnostic build (A/B): every engine field loaded at kernel entry, as before round 3
        const DevEngine *Ec = &E;
#else
        uint32_t koff = 0;
        asm volatile("" : "+s"(koff));
        const DevEngine *Ec = (const DevEngine *)((const __attribute__((address_space(4))) char *)
                                                  __builtin_amdgcn_kernarg_segment_ptr() + koff);
#endif
        TP(1, choose_body<SLOTS>(*Ec))
        wave_mem_sync();
        TP(2, advance_body<SLOTS>(*Ec, nullptr, 1))
        wave_mem_sync();
    }
#ifdef AZX_STAMP_PLAY
    if (threadIdx.x == 0)
        for (int i = 0; i < 3; ++i) E.counters[(size_t)blockIdx.x * CTR_COUNT + 10 + i] += tp_[i];
#endif
#undef TP
}

template <int SLOTS>
__global__ __launch_bounds__(64) void k_hex_replay(int N, int n_games, const int32_t *moves,
                                                   const int32_t *length, int stride,
                                                   int32_t *result_out, int32_t *nlegal_out,
                                                   uint64_t *empties_out, int32_t *final_board) {
    const int lane = threadIdx.x;
    const int g = blockIdx.x;
    if (g >= n_games) return;
    const int ncells = N * N;
    HexWave<SLOTS> h;
    h.clear();
    h.geom(N, lane);
    const int len = length[g];
    for (int p = 0; p < len; ++p) {
        const Masks<SLOTS> mk = make_masks<SLOTS>(h, lane, ncells);
        const size_t o = (size_t)g * stride + p;
        if (lane == 0) {
            nlegal_out[o] = mk.k;
#pragma unroll
            for (int s = 0; s < 4; ++s) empties_out[o * 4 + s] = (s < SLOTS && mk.k) ? mk.m[s < SLOTS ? s : 0] : 0ull;
        }
        h.step(moves[o] - 1, N, lane);
        if (lane == 0) result_out[o] = h.winner ? (h.winner == 2 ? 1 : 3) : 0;   // hex.py:161-170
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int cell = s * 64 + lane;
        if (cell < ncells) final_board[(size_t)g * ncells + cell] = (int32_t)(h.c[s] & 3u);
    }
}

// ---- float32 arithmetic self-test: sqrt / divide must be IEEE correctly rounded -------------
__global__ void k_arith(const float *a, const float *b, float *sq, float *dv, float *mul, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        sq[i] = sqrtf(a[i]);
        dv[i] = a[i] / (1.0f + b[i]);
        mul[i] = (0.75f * a[i]) * b[i] + a[i];
    }
}

// ---- the score path's shortcuts against their IEEE definitions: q = num/den through
// div2_unscaled (pairs of elements, as the kernel uses it) and rt = the sqrt table entry of the
// integer den (0 beyond the table) ----------------------------------------------------------------
__global__ void k_divide_test(const float *num, const float *den, float *q, float *rt, int n) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 < n) {
        const f2 a = {num[i], num[i + 1]}, d = {den[i], den[i + 1]};
        const f2 r = div2_unscaled(a, d);
        q[i] = r.x;
        q[i + 1] = r.y;
    } else if (i < n) {
        const f2 a = {num[i], num[i]}, d = {den[i], den[i]};
        q[i] = div2_unscaled(a, d).x;
    }
    for (int j = i; j < n && j < i + 2; ++j) {
        const int v = (int)den[j];
        rt[j] = (v >= 0 && v < AZX_SQRT_TAB) ? c_sqrt[v] : 0.0f;
    }
}

// ---- device Dirichlet self-test: rows of k-cell noise exactly as k_mcts draws it ---------------
__global__ __launch_bounds__(64) void k_noise_test(float alpha, const float *tab, int k, int n_rows, uint32_t seed, float *out) {
    const int lane = threadIdx.x, row = blockIdx.x;
    if (row >= n_rows) return;
    uint64_t m[2];
    m[0] = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    m[1] = k > 64 ? ((1ull << (k - 64)) - 1ull) : 0ull;
    const GammaConst gc = gamma_const(alpha, tab);
    float nz[2];
    dirichlet_noise<2>(m, lane, mix32(seed ^ mix32((uint32_t)row + 0x632be5abu)), gc, 1.0f, nz);
    if (lane < k) out[(size_t)row * k + lane] = nz[0];
    if (64 + lane < k) out[(size_t)row * k + 64 + lane] = nz[1];
}

void azx_launch_noise_test(float alpha, const float *tab, int k, int n_rows, uint32_t seed, float *out, hipStream_t st) {
    hipLaunchKernelGGL(k_noise_test, dim3(n_rows), dim3(64), 0, st, alpha, tab, k, n_rows, seed, out);
}

// ---- host: the delta table of log2_gamma_variate for one alpha (double precision) ---------------
namespace {
// ln of the regularised lower / upper incomplete gamma function P(a,x), Q(a,x)
void ln_inc_gamma(double a, double x, double *lnP, double *lnQ) {
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {                       // series for P
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 2000; ++n) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
        }
        *lnP = -x + a * std::log(x) + std::log(sum) - lg;
        const double P = std::exp(*lnP);
        *lnQ = P < 1.0 ? std::log1p(-P) : -1e300;
    } else {                                 // continued fraction for Q (modified Lentz)
        const double tiny = 1e-300;
        double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
        for (int i = 1; i < 5000; ++i) {
            const double an = -i * (i - a);
            b += 2.0;
            d = an * d + b;
            if (std::fabs(d) < tiny) d = tiny;
            c = b + an / c;
            if (std::fabs(c) < tiny) c = tiny;
            d = 1.0 / d;
            const double del = d * c;
            h *= del;
            if (std::fabs(del - 1.0) < 1e-16) break;
        }
        *lnQ = -x + a * std::log(x) + std::log(h) - lg;
        const double Q = std::exp(*lnQ);
        *lnP = Q < 1.0 ? std::log1p(-Q) : -1e300;
    }
}
// ln x with P(a,x) = 1 - y, by bisection on ln x (matching ln P when y > 1/2, ln Q otherwise)
double ln_gamma_quantile_from_y(double a, double y) {
    const bool lower = y > 0.5;
    const double target = lower ? std::log1p(-y) : std::log(y);
    double lo = -800.0 / a - 50.0, hi = std::log(80.0);
    if (lower) {   // tiny x: ln P = a ln x - lgamma(a+1) + O(x); start near there
        const double est = (target + std::lgamma(a + 1.0)) / a;
        lo = est - 2.0;
        hi = std::min(hi, est + 40.0);
    }
    for (int it = 0; it < 200; ++it) {
        const double mid = 0.5 * (lo + hi);
        double lp, lq;
        const double x = std::exp(mid);
        if (x <= 0.0) { lo = mid; continue; }
        ln_inc_gamma(a, x, &lp, &lq);
        const bool below = lower ? (lp < target) : (lq > target);   // P grows, Q falls with x
        if (below) lo = mid; else hi = mid;
    }
    return 0.5 * (lo + hi);
}
}  // namespace

void azx_gamma_table(double alpha, float *tab) {
    const double ln2 = std::log(2.0);
    const double c0 = std::lgamma(alpha + 1.0) / ln2;            // log2 Gamma(alpha + 1)
    for (int j = 0; j < AZX_GAMMA_TAB; ++j) {
        const int e = (j >> 5) - 25;                              // y = 2^e * (1 + (j & 31) / 32)
        const double y = std::ldexp(1.0 + (j & 31) / 32.0, e);
        double delta = 0.0;
        if (y < 1.0) {
            const double u = 1.0 - y;
            const double lx0 = (std::log2(u) + c0) / alpha;
            if (lx0 > -60.0) delta = ln_gamma_quantile_from_y(alpha, y) / ln2 - lx0;   // below: x^alpha form is exact
        }
        tab[j] = (float)delta;
    }
    tab[AZX_GAMMA_TAB] = (float)c0;
}

// ============================================================================================
// host-side launchers (called from azx_capi.cpp)
// ============================================================================================
#define DISPATCH_SLOTS(slots, CALL)                         \
    do {                                                    \
        if ((slots) <= 2) { CALL(2); }                      \
        else { CALL(3); }                                   \
    } while (0)

// force_generic (azx_create reads AZX_MCTS_GENERIC once per engine) keeps every launch on the generic
// instantiation (A/B and the equivalence test)
bool azx_mcts_fast_path(const DevEngine &E, int mode, bool force_generic) {
    return !force_generic && mode == (MODE_BEGIN | MODE_INLINE) && E.evaluator == AZX_EVAL_UNIFORM &&
           E.prior_default && (E.noise_scale == 0.0 || E.device_noise);
}

void azx_launch_mcts(const DevEngine &E, int mode, int num_batches, hipStream_t st, bool force_generic) {
    const size_t lds = azx_mcts_lds_bytes(E.ncells, E.bs);
    if (azx_mcts_fast_path(E, mode, force_generic)) {
#define CALL(S) hipLaunchKernelGGL((k_mcts<S, true>), dim3(E.G), dim3(64), lds, st, E, mode, num_batches)
        DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
        return;
    }
#define CALL(S) hipLaunchKernelGGL((k_mcts<S, false>), dim3(E.G), dim3(64), lds, st, E, mode, num_batches)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
}

// FAST conditions as in azx_launch_mcts; returns false (nothing launched) when they do not hold
bool azx_launch_play(const DevEngine &E, int num_batches, int steps, hipStream_t st, bool allowed) {
    if (!allowed) return false;
    if (!(E.evaluator == AZX_EVAL_UNIFORM && E.prior_default && (E.noise_scale == 0.0 || E.device_noise))) return false;
    const size_t lds = azx_mcts_lds_bytes(E.ncells, E.bs);
#define CALL(S) hipLaunchKernelGGL((k_play<S>), dim3(E.G), dim3(64), lds, st, E, num_batches, steps)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
    return true;
}

void azx_launch_reset(const DevEngine &E, const int32_t *slots, int n_slots, const int32_t *moves,
                      const int32_t *n_moves, int stride, int assign_uid, hipStream_t st) {
#define CALL(S) hipLaunchKernelGGL((k_reset<S>), dim3(n_slots), dim3(64), 0, st, E, slots, n_slots, moves, n_moves, stride, assign_uid)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
}

void azx_launch_advance(const DevEngine &E, const int32_t *move_ids, int play_mode, hipStream_t st) {
#define CALL(S) hipLaunchKernelGGL((k_advance<S>), dim3(E.G), dim3(64), 0, st, E, move_ids, play_mode)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
}

void azx_launch_gather_root(const DevEngine &E, int32_t *k_out, int32_t *legal, float *cv, float *cw,
                            float *cp, float *rv, float *rw, int32_t *nn, float *sv, hipStream_t st) {
#define CALL(S) hipLaunchKernelGGL((k_gather_root<S>), dim3(E.G), dim3(64), 0, st, E, k_out, legal, cv, cw, cp, rv, rw, nn, sv)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
}

void azx_launch_choose(const DevEngine &E, hipStream_t st) {
#define CALL(S) hipLaunchKernelGGL((k_choose<S>), dim3(E.G), dim3(64), 0, st, E)
    DISPATCH_SLOTS(E.slots, CALL);
#undef CALL
}

void azx_launch_hex_replay(int N, int n_games, const int32_t *moves, const int32_t *length, int stride,
                           int32_t *result_out, int32_t *nlegal_out, uint64_t *empties_out,
                           int32_t *final_board, hipStream_t st) {
    const int slots = (N * N + 63) / 64;
#define CALL(S) hipLaunchKernelGGL((k_hex_replay<S>), dim3(n_games), dim3(64), 0, st, N, n_games, moves, length, stride, result_out, nlegal_out, empties_out, final_board)
    DISPATCH_SLOTS(slots, CALL);
#undef CALL
}

void azx_launch_divide_test(const float *num, const float *den, float *q, float *rt, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_divide_test, dim3((n / 2 + 256) / 256), dim3(256), 0, st, num, den, q, rt, n);
}

void azx_launch_arith(const float *a, const float *b, float *sq, float *dv, float *mul, int n,
                      hipStream_t st) {
    hipLaunchKernelGGL(k_arith, dim3((n + 255) / 256), dim3(256), 0, st, a, b, sq, dv, mul, n);
}
