// Decoder hot path for gfx950: clamped sigmoid + 5x5 NMS + peak compaction, exact top-k
// selection, offset/embedding gather and anchor<->part association.
// Follows src/sdnet/data/decoders.py:41-100 and src/sdnet/utils/utils.py:341-361,422-467 of
// the reference.  All arithmetic that decides an index is done exactly as the reference's fp32
// tensor ops: separately rounded mul/add/sqrt -- floating-point contraction is OFF in this
// file.
#pragma clang fp contract(off)
#include "sd_common.h"
#include <atomic>
#include <cstring>

namespace sd {

// ---------------------------------------------------------------------------------------------
// keys: 64-bit, larger = better.  high word = order-preserving transform of the fp32 score,
// low word = ~flat (flat = class*h*w + y*w + x) so that among equal scores the lower class /
// lower flat index wins (stable order of the class-major flattened map).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f2ord(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}
__device__ __forceinline__ uint64_t make_key(float v, uint32_t flat) {
    return ((uint64_t)f2ord(v) << 32) | (uint64_t)(~flat);
}

#ifdef SD_DECODE_TRACE
// timing experiment (not in product builds; reported by sd_build_flags bit 3): 100 MHz timestamps of the stages of k_decode_fused
__device__ unsigned long long sd_trace[8192];
#define SD_TRACE(slot) do { if (threadIdx.x == 0 && (slot) >= 0 && (slot) < 8192) sd_trace[(slot)] = wall_clock64(); } while (0)
#else
#define SD_TRACE(slot) do { } while (0)
#endif
// ---------------------------------------------------------------------------------------------
// Kernel 1: tile NMS.  One 256-thread block per 64x16 output tile of one map; the (64+4)x(16+4)
// neighbourhood is staged once in LDS as clamped-sigmoid values, 5-max is separable
// (row pass into a second LDS array, column pass in registers).  HBM-bound: every logit is read
// once (+ halo re-reads, 1.33x, served by L2).
//   MODE 0: dense output  out = keep ? v : 0            (nms(), utils.py:441-443)
//   MODE 1: compaction    survivors appended as keys to the per-(image, group) candidate list
// ---------------------------------------------------------------------------------------------
#ifndef SD_DECODE_TH
#define SD_DECODE_TH 16      // NMS tile height (A/B switch: 32 halves the number of tile blocks)
#endif
constexpr int TW = 64, TH = SD_DECODE_TH, HALO = 2;
constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;

struct Group {
    const float* p;
    int64_t sb, sc;
    int C;
};

// Candidate counters live one per 128-byte line: every tile block bumps its (image, group) counter once, and at 16384 blocks
// (stress config) 32 adjacent ints in one line serialised all of them in one L2 channel (measured 151 us for a 67 MB read).
constexpr int CNT_STRIDE = 32;

template <int MODE>
__global__ __launch_bounds__(256) void k_nms_tile(Group g0, Group g1, int h, int w, int tiles_x, int apply_sigmoid, float min_score,
                                                   float* __restrict__ dense_out,    // MODE 0: (B, C0, h, w)
                                                   uint64_t* __restrict__ cand0, uint64_t* __restrict__ cand1,
                                                   int* __restrict__ counters) {     // MODE 1: counters[(b*2+g) * CNT_STRIDE]
    __shared__ float S[LH][LW];
    __shared__ float Hm[LH][TW];
    __shared__ uint64_t keep_keys[TW * TH];
    __shared__ int keep_n, keep_base;

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    int c = blockIdx.y;
    const int grp = (c >= g0.C) ? 1 : 0;
    const Group g = grp ? g1 : g0;
    if (grp) c -= g0.C;
    const int tx0 = (blockIdx.x % tiles_x) * TW;
    const int ty0 = (blockIdx.x / tiles_x) * TH;
    const float* plane = g.p + (int64_t)b * g.sb + (int64_t)c * g.sc;

    if (tid == 0) keep_n = 0;
    // all loads of the thread are issued before the first use (a loop with the bounds test around the load is not pipelined
    // by hipcc: six dependent L2 round trips per block); out-of-image cells read element 0 and are replaced by -inf
    constexpr int NLD = (LH * LW + 255) / 256;
    float ld[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + j * 256;
        const int r = i / LW, cc = i - r * LW;
        const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
        const bool ok = i < LH * LW && y >= 0 && y < h && x >= 0 && x < w;
        ld[j] = plane[ok ? (int64_t)y * w + x : 0];
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + j * 256;
        const int r = i / LW, cc = i - r * LW;
        const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
        const bool ok = y >= 0 && y < h && x >= 0 && x < w;
        if (i < LH * LW) S[r][cc] = ok ? (apply_sigmoid ? clamped_sigmoid(ld[j]) : ld[j]) : -INFINITY;
    }
    __syncthreads();
    for (int i = tid; i < LH * TW; i += 256) {
        const int r = i / TW, cc = i - r * TW;
        float m = fmaxf(fmaxf(S[r][cc], S[r][cc + 1]), fmaxf(S[r][cc + 2], S[r][cc + 3]));
        Hm[r][cc] = fmaxf(m, S[r][cc + 4]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (TW * TH) / 256; ++j) {
        const int i = tid + j * 256;
        const int r = i / TW, cc = i - r * TW;
        const int y = ty0 + r, x = tx0 + cc;
        float m = fmaxf(fmaxf(Hm[r][cc], Hm[r + 1][cc]), fmaxf(Hm[r + 2][cc], Hm[r + 3][cc]));
        m = fmaxf(m, Hm[r + 4][cc]);
        const float v = S[r + HALO][cc + HALO];
        const bool inside = (y < h) && (x < w);
        // min_score = fp32(conf) in the annotations-only mode: `>=` keeps a score EQUAL to fp32(conf), which the reference still
        // emits as a part-less object when double(score) > conf (decoders.py:115-117, e.g. conf = 0.4 -> fp32 0.4000000060);
        // the fp32 `score > conf` mask of the association stage and the host's double compare decide from there
        const bool keep = inside && (v == m) && (MODE == 0 || v >= min_score);
        if (MODE == 0) {
            if (inside) dense_out[(((int64_t)b * g0.C + c) * h + y) * w + x] = keep ? v : 0.0f;
        } else if (keep) {
            const int slot = atomicAdd(&keep_n, 1);
            keep_keys[slot] = make_key(v, (uint32_t)(c * h * w + y * w + x));
        }
    }
    if (MODE == 1) {
        __syncthreads();
        const int n = keep_n;
        if (n == 0) return;
        if (tid == 0) keep_base = atomicAdd(&counters[(b * 2 + grp) * CNT_STRIDE], n);
        __syncthreads();
        const int64_t cap = (int64_t)g.C * h * w;
        uint64_t* dst = (grp ? cand1 : cand0) + (int64_t)b * cap + keep_base;
        for (int i = tid; i < n; i += 256) dst[i] = keep_keys[i];
    }
}

// dense map -> keys (generic topk(), utils.py:447-467: every pixel is a candidate)
__global__ __launch_bounds__(256) void k_dense_keys(const float* __restrict__ scores, int64_t sb, int64_t sc, int C, int hw,
                                                     uint64_t* __restrict__ cand) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hw) return;
    const float v = scores[(int64_t)b * sb + (int64_t)c * sc + i];
    cand[((int64_t)b * C + c) * hw + i] = make_key(v, (uint32_t)(c * hw + i));
}

// ---------------------------------------------------------------------------------------------
// Block-wide exact top-k of n unique 64-bit keys (descending) into LDS `buf[0..k)`.
//   n <= SORT_CAP : load everything into LDS, bitonic sort.
//   else          : MSB-first 8-bit radix select on the global list to find the k-th largest
//                   key, collect the k keys >= it, bitonic sort those.
// Slots beyond min(n,k) are left as key 0 (filled by the caller).
// ---------------------------------------------------------------------------------------------
constexpr int SEL_THREADS = 512;      // threads of one selection "team" (8 waves) in the two-launch kernels
constexpr int SORT_CAP = 4096;

// Candidate keys written by OTHER workgroups of the SAME launch (k_decode_fused) are read with agent-scope relaxed atomic
// loads (global_load_dwordx2 sc1: bypasses this CU's L1, which is never refreshed by another CU's stores -- guide, Guideline 16);
// lists written by an earlier launch are read with plain loads.
template <bool COH>
__device__ __forceinline__ uint64_t ldkey(const uint64_t* p) {
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// One team = SEL_THREADS threads working on one candidate list.  k_select_group runs two teams in one
// 1024-thread block (anchors and parts side by side): every team executes the SAME sequence of block barriers
// (the sequence depends only on `np2` / `use_radix`, which the caller makes identical for both teams).
struct Team {
    int tid;            // 0 .. SEL_THREADS-1 inside the team
    uint64_t* buf;      // [SORT_CAP]
    int* hist;          // [2][256]  (double-buffered by radix pass)
    int* misc;          // [4]
    int* flags;         // [SD_MAX_TOPK]
    uint64_t* out;      // [out_keys(k_max)]: the selected keys of an LDS-resident radix select before they replace buf[0..k)
    int team;           // index of this team inside the block
    int* alive;         // [2], shared by ALL teams of the block: 1 while a team still needs radix passes
};

// In-place descending bitonic sort of buf[0..np2).  Each wave owns a contiguous range of R elements; stages
// whose compare distance stays inside a range need no block barrier (LDS operations of one wave execute in
// order), only the few long-distance stages synchronise the whole block: 6 instead of 66 barriers for 2048 keys.
template <int NT>
__device__ void bitonic_desc(const Team& T, uint64_t* buf, int np2) {
    constexpr int WAVES = NT / 64;
    constexpr int NW = WAVES >= 16 ? 16 : (WAVES >= 8 ? 8 : (WAVES >= 4 ? 4 : (WAVES >= 2 ? 2 : 1)));   // ranges: a power of two (192-thread blocks: two)
    const int lane = T.tid & 63, wave = T.tid >> 6;
    const int R = max(np2 / NW, 128);                 // elements per wave range
    const int half_pairs = min(R, np2) >> 1;          // compare-exchange pairs per range and stage
    const bool active = wave < NW && wave * R < np2;
    bool local_dirty = false;                         // wave-local stages since the last block barrier
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (2 * j <= R) {                         // partner inside the wave's own range
                if (active) {
                    const int base = wave * R;
                    for (int t = lane; t < half_pairs; t += 64) {
                        const int i = base + 2 * t - (t & (j - 1)), l = i + j;
                        const uint64_t a = buf[i], bb = buf[l];
                        if ((a < bb) == ((i & k) == 0)) { buf[i] = bb; buf[l] = a; }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-level ordering of the LDS traffic
                local_dirty = true;
            } else {
                if (local_dirty) { __syncthreads(); local_dirty = false; }
                for (int t = T.tid; t < (np2 >> 1); t += NT) {
                    const int i = 2 * t - (t & (j - 1)), l = i + j;
                    const uint64_t a = buf[i], bb = buf[l];
                    if ((a < bb) == ((i & k) == 0)) { buf[i] = bb; buf[l] = a; }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}

// Descending sort of buf[0..np2) by ranking (np2 <= RANK_CAP, buf holds >= 2 * np2 keys): every thread counts the keys that
// precede its own (LDS broadcast reads, no barrier inside) and drops it at that rank.  Two block barriers -- the bitonic network
// needs 15-28 dependent LDS round trips for 32-128 keys (3-5 us measured in the one-launch decoder), this one pass.
// Equal keys (only the zero padding) are ordered by index.
constexpr int RANK_CAP = 256;
template <int NT>
__device__ void rank_sort_desc(const Team& T, uint64_t* buf, int np2) {
    if (np2 * 4 <= NT && np2 >= 16) {                        // four lanes per key, each counting over a quarter of the list (np2 is team-uniform)
        const int i = T.tid >> 2, q = T.tid & 3;
        const bool in = i < np2;
        const uint64_t key = in ? buf[i] : 0ull;
        int rank = 0;
        if (in) {
#pragma unroll 4
            for (int j = q * 2; j < np2; j += 8) {
                const uint64_t a = buf[j], bb = buf[j + 1];
                rank += (a > key || (a == key && j < i)) ? 1 : 0;
                rank += (bb > key || (bb == key && j + 1 < i)) ? 1 : 0;
            }
        }
        rank += __shfl_xor(rank, 1);
        rank += __shfl_xor(rank, 2);
        if (in && q == 0) buf[np2 + rank] = key;
        __syncthreads();
        for (int t = T.tid; t < np2; t += NT) buf[t] = buf[np2 + t];
        __syncthreads();
        return;
    }
    for (int i = T.tid; i < np2; i += NT) {
        const uint64_t key = buf[i];
        int rank = 0;
#pragma unroll 4
        for (int j = 0; j < np2; j += 2) {
            const uint64_t a = buf[j], bb = buf[j + 1];
            rank += (a > key || (a == key && j < i)) ? 1 : 0;
            rank += (bb > key || (bb == key && j + 1 < i)) ? 1 : 0;
        }
        buf[np2 + rank] = key;
    }
    __syncthreads();
    for (int i = T.tid; i < np2; i += NT) buf[i] = buf[np2 + i];
    __syncthreads();
}

// Descending bitonic sort of buf[0..np2), np2 <= NT, ONE key per thread held in a register: the 64-lane stages exchange through
// cross-lane shuffles, only compare distances >= 64 go through LDS (two block barriers each).  The LDS network above needs a dependent
// LDS round trip + wait per stage (45 stages for 512 keys: 9.9 us measured in the per-image merge); here 512 keys take 39 shuffle
// stages + 6 LDS stages.  Same compare-exchange network, same result.  Every thread of the block runs the same barrier sequence
// (np2 must be block-uniform: teams pass the common size).
template <int NT>
__device__ void reg_bitonic_desc(const Team& T, uint64_t* buf, int np2) {
    const int i = T.tid;
    const bool in = i < np2;
    uint64_t key = in ? buf[i] : 0ull;
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            uint64_t other;
            if (j < 64) {
                other = __shfl_xor((unsigned long long)key, j);
            } else {
                __syncthreads();                          // everybody has read the previous exchange
                if (in) buf[i] = key;
                __syncthreads();
                other = in ? buf[i ^ j] : 0ull;
            }
            const bool lower = (i & j) == 0;              // this thread holds the lower index of the pair (i, i ^ j)
            const uint64_t lo = lower ? key : other, hi = lower ? other : key;
            const int li = lower ? i : (i ^ j);
            const bool swap = (lo < hi) == ((li & k) == 0);
            key = (lower == swap) ? hi : lo;              // lower index takes hi when swapped, upper takes lo
        }
    }
    __syncthreads();
    if (in) buf[i] = key;
    __syncthreads();
}

// keys the `out` buffer of a team must hold for selections of up to k_max keys
__host__ __device__ constexpr int out_keys(int np2k) { return np2k <= RANK_CAP ? 2 * np2k : np2k; }

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Key sources of the selection.  FlatSrc: one contiguous list (k_nms_tile<1>'s append buffer).  TiledSrc: one fixed region of
// TILE_CAP slots per NMS tile plus per-tile counts (k_decode_fused: no global append counter on the tile blocks' critical path).
template <int NT, bool COH>
struct FlatSrc {
    const uint64_t* p;
    int n;
    template <class F>
    __device__ __forceinline__ void for_each(int tid, F f) const {          // key loads batched four deep
        for (int base = tid; base < n; base += 4 * NT) {
            uint64_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) key[u] = (base + u * NT < n) ? ldkey<COH>(p + base + u * NT) : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * NT < n) f(key[u]);
        }
    }
    __device__ __forceinline__ void fill(int tid, uint64_t* buf, int np2) const {
        for (int i = tid; i < np2; i += NT) buf[i] = (i < n) ? ldkey<COH>(p + i) : 0ull;
    }
};

// (a tile owns TW x tile-height candidate slots -- TILE_CAP in the comments: every pixel of a plateau survives the NMS)
constexpr int SPEC = 6;               // keys of a tile that travel INSIDE its 64-byte hand-off record (count word + 6 keys): one round trip
constexpr int SPEC_TILES = 256;       // ... captured in LDS for the first SPEC_TILES tiles of the image (the others re-read them from the slots)

template <int NT, bool COH = true>
struct TiledSrc {
    const uint64_t* base;   // slot 0 of the list's first tile (global; COH: written by other workgroups of this launch -> sc1 loads)
    const int* cnt;         // LDS: candidates of each tile of the list
    const int* off;         // LDS: exclusive prefix of cnt within the list
    const uint64_t* spec;   // LDS: the first SPEC keys of the tiles that were loaded speculatively (image-wide tile index)
    int tile0;              // image-wide index of the list's first tile
    int ntiles, n;
    int cap;                // candidate slots of one tile (TW x tile height of the launch)
    __device__ __forceinline__ uint64_t key_at(int t, int j) const {
        const int ti = tile0 + t;
        if (j < SPEC && ti < SPEC_TILES) return spec[ti * SPEC + j];
        return ldkey<COH>(base + (int64_t)t * cap + j);
    }
    // candidate g of the list (0 <= g < n) -> (tile, slot): binary search in the exclusive prefix `off`
    __device__ __forceinline__ uint64_t key_of(int gidx) const {
        int lo = 0, hi = ntiles - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (off[mid] <= gidx) lo = mid; else hi = mid - 1;
        }
        return key_at(lo, gidx - off[lo]);
    }
    template <class F>
    __device__ __forceinline__ void for_each(int tid, F f) const {          // all threads busy, loads batched four deep
        for (int base = tid; base < n; base += 4 * NT) {
            uint64_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) key[u] = (base + u * NT < n) ? key_of(base + u * NT) : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * NT < n) f(key[u]);
        }
    }
    __device__ __forceinline__ void fill(int tid, uint64_t* buf, int np2) const {
        for (int i = n + tid; i < np2; i += NT) buf[i] = 0ull;
        for (int base = tid; base < n; base += 4 * NT) {
            uint64_t key[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) key[u] = (base + u * NT < n) ? key_of(base + u * NT) : 0ull;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (base + u * NT < n) buf[base + u * NT] = key[u];
        }
    }
};

// keys already resident in LDS (second stage of the selection)
template <int NT>
struct LdsSrc {
    const uint64_t* p;
    int n;
    template <class F>
    __device__ __forceinline__ void for_each(int tid, F f) const {
        for (int i = tid; i < n; i += NT) f(p[i]);
    }
};

// Histogram increment with wave-level pre-aggregation.  NMS survivors are clamped sigmoids of a narrow score range: in the first radix
// pass nearly all keys of a list share ONE digit (sign + exponent byte), and 64 lanes adding to one LDS word are 64 serialised atomics
// per wave instruction (measured: ~20 us for a per-map selection of 1.4 k keys).  Two rounds of "the first active lane's digit: one
// lane adds the population count of its group" take the dominant digits out; whatever is left (spread digits: distinct LDS words,
// conflict-free) goes through plain atomics.  Called under divergent control flow: ballots see the active lanes only.
__device__ __forceinline__ void hist_add(int* h, int digit) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(1);
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        if (todo == 0ull) return;
        const int leader = __ffsll((long long)todo) - 1;
        const int d = __builtin_amdgcn_readlane(digit, leader);       // (leader is wave-uniform: v_readlane, not a trip through the LDS crossbar)
        const unsigned long long same = __ballot(digit == d) & todo;
        if (lane == leader) atomicAdd(&h[d], __popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&h[digit], 1);
}

// Slot allocation from one LDS counter for the lanes with `take` set: one atomic per wave instead of one per lane.
__device__ __forceinline__ int alloc_slot(int* counter, bool take) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(take);
    if (m == 0ull) return -1;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __builtin_amdgcn_readlane(base, leader);
    return take ? base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) : -1;
}

// Block barrier that publishes LDS data only.  __syncthreads() is a workgroup-scope fence + s_barrier: it also waits for the wave's global
// STORES in flight (s_waitcnt vmcnt(0): ~1.5 us for a store's acknowledgement), which nobody in the block reads -- where the anchors' outputs
// are stored just before the barrier that hands their positions to the parts' threads, every thread of the block waited for those stores
// (traced: 3.3 us from the zero slots to the end of k_rank_group_small, 2 of them behind this barrier).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Block-wide sum / maximum into one LDS word with ONE atomic per wave: lanes of a wave adding to the same LDS address serialise (~20 cycles
// each -- k_group_wide spent 3.6 us of 13.5 on two counters every lane added to).  Call from converged code (every lane of the wave).
__device__ __forceinline__ void wave_atomic_add(int* counter, int v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(counter, v);
}
__device__ __forceinline__ void wave_atomic_max(int* counter, int v) {             // (v >= 0; *counter starts at 0)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v = max(v, __shfl_xor(v, o));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(counter, v);
}

// MSB-first 8-bit radix select of the k largest of the source's n unique keys into dst[0..np2k) (zero padded), then sorted
// descending.  Two block barriers per pass (histograms double-buffered by pass parity, bucket scan by one wave); the passes stop
// as soon as the boundary bucket is taken whole in EVERY team of the block (unique keys: usually after the score bytes) --
// `alive` is block-wide so that all teams leave together.  dst holds out_keys(np2k) keys and must not alias the source.
template <int NT, class Src>
__device__ void radix_select_sorted(const Team& T, const Src& src, int k, uint64_t* dst, int np2k) {
    const int tid = T.tid;
    const int n = src.n;
    uint64_t prefix = 0, mask = 0;
    int remaining = min(k, n);
    bool done = remaining == 0;
    if (done) prefix = ~0ull;
    for (int i = tid; i < 512; i += NT) T.hist[i] = 0;          // double-buffered histogram
    if (tid == 0) T.alive[T.team] = done ? 0 : 1;
    __syncthreads();
    [[maybe_unused]] const int rtrace = (blockIdx.x == 0 && T.team == 0) ? 6300 : -100;       // (trace builds: radix select of block 0, team 0)
    SD_TRACE(rtrace + 0);
    for (int pass = 7; pass >= 0; --pass) {
        int* hcur = T.hist + (pass & 1) * 256;
        int* hnext = T.hist + ((pass & 1) ^ 1) * 256;
        const int shift = pass * 8;
        if (!done)
            src.for_each(tid, [&](uint64_t key) {
                if ((key & mask) == prefix) hist_add(hcur, (int)((key >> shift) & 255ull));
            });
        for (int i = tid; i < 256; i += NT) hnext[i] = 0;
        __syncthreads();
        if (!done && tid < 64) {          // one wave: lane l owns digits 255-4l .. 252-4l (descending)
            const int c0 = hcur[255 - 4 * tid], c1 = hcur[254 - 4 * tid], c2 = hcur[253 - 4 * tid], c3 = hcur[252 - 4 * tid];
            const int sum = c0 + c1 + c2 + c3;
            int incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (tid >= o) incl += v;
            }
            const int excl = incl - sum;
            if (excl < remaining && remaining <= incl) {      // the lane whose digits hold the k-th key
                int acc = excl, d = 255 - 4 * tid, cnt = c0;
                if (acc + c0 < remaining) { acc += c0; --d; cnt = c1;
                    if (acc + c1 < remaining) { acc += c1; --d; cnt = c2;
                        if (acc + c2 < remaining) { acc += c2; --d; cnt = c3; } } }
                T.misc[0] = d;
                T.misc[1] = remaining - acc;                  // keys to take inside digit d
                T.alive[T.team] = (cnt == remaining - acc || pass == 0) ? 0 : 1;   // whole bucket taken: threshold known
            }
        }
        __syncthreads();
        if (!done) {
            prefix |= (uint64_t)T.misc[0] << shift;
            mask |= 255ull << shift;
            remaining = T.misc[1];
            done = T.alive[T.team] == 0;
        }
        SD_TRACE(rtrace + 8 - pass);
        if ((T.alive[0] | T.alive[1]) == 0) break;
    }
    SD_TRACE(rtrace + 10);
    // keys are unique, so exactly min(k, n) keys are >= prefix
    for (int i = tid; i < np2k; i += NT) dst[i] = 0ull;
    if (tid == 0) T.misc[2] = 0;
    __syncthreads();
    src.for_each(tid, [&](uint64_t key) {
        const int slot = alloc_slot(&T.misc[2], key >= prefix);
        if (slot >= 0 && slot < np2k) dst[slot] = key;
    });
    __syncthreads();
    SD_TRACE(rtrace + 11);
    if (np2k <= RANK_CAP) rank_sort_desc<NT>(T, dst, np2k);
    else if (np2k <= NT) reg_bitonic_desc<NT>(T, dst, np2k);
    else bitonic_desc<NT>(T, dst, np2k);
    SD_TRACE(rtrace + 12);
}

// Exact top-k of the source's n unique keys (descending) into T.buf[0..k).  The path depends only on (n_max, k_max, cap), the
// longest list / largest k of ALL teams of the block and the key capacity of T.buf, so every team runs the same barriers:
//   A  everything fits the ranking sort (<= RANK_CAP keys): load, rank, done -- the annotations-only mode (tens of peaks);
//   B  the list fits T.buf: load it ONCE into LDS, radix-select the k largest from LDS into T.out, sort those, move them to
//      T.buf[0..k) -- a bitonic sort of 2048 keys took 30 us per image (exact top-k at 512x512), this takes a tenth;
//   C  longer lists: the same radix select reading the source (global memory) once per pass, selected keys straight into T.buf.
// Slots beyond min(n, k) are left as key 0 (filled by the caller).
template <int NT, class Src>
__device__ void team_select_topk(const Team& T, const Src& src, int k, int n_max, int k_max, int cap) {
    const int tid = T.tid;
    const int np2_all = max(next_pow2(max(n_max, k_max)), 2);
    if (np2_all <= RANK_CAP) {
        src.fill(tid, T.buf, np2_all);
        __syncthreads();
        rank_sort_desc<NT>(T, T.buf, np2_all);
        return;
    }
    const int np2k = max(next_pow2(k_max), 2);
    if (n_max <= cap) {
        src.fill(tid, T.buf, src.n);
        __syncthreads();
        radix_select_sorted<NT>(T, LdsSrc<NT>{T.buf, src.n}, k, T.out, np2k);
        for (int i = tid; i < k; i += NT) T.buf[i] = T.out[i];
        __syncthreads();
        return;
    }
    radix_select_sorted<NT>(T, src, k, T.buf, np2k);
}

// Suppressed pixels have score exactly 0; when fewer than k peaks exist the reference's remaining top-k slots
// are zeros (utils.py:451 on the NMS'ed map).  Fill them with the lowest class-major flat indices that are not
// peaks (stable order).  Always executes two block barriers (team-uniform control flow).
template <int NT>
__device__ void fill_zero_slots(const Team& T, int npos, int k) {
    const int tid = T.tid;
    if (k <= 64) {                                    // one wave: unused flat indices ranked with a ballot (the default K = 20 / P = 40)
        if (npos < k && tid < 64) {
            int used = 0;
            for (int j = 0; j < npos; ++j) used |= ((uint32_t)(~T.buf[j]) == (uint32_t)tid);
            const bool flag = tid < k && !used;
            const unsigned long long m = __ballot(flag);
            const int rank = __popcll(m & ((1ull << tid) - 1ull));
            if (flag && npos + rank < k) T.buf[npos + rank] = make_key(0.0f, (uint32_t)tid);
        }
        __syncthreads();
        __syncthreads();                              // (same barrier count as the general path: teams may take different paths)
        return;
    }
    if (npos < k) {
        for (int f = tid; f < k; f += NT) {
            int used = 0;
            for (int j = 0; j < npos; ++j) used |= ((uint32_t)(~T.buf[j]) == (uint32_t)f);
            T.flags[f] = used ? 0 : 1;
        }
    }
    __syncthreads();
    if (npos < k) {
        for (int f = tid; f < k; f += NT) {
            if (!T.flags[f]) continue;
            int rank = 0;
            for (int j = 0; j < f; ++j) rank += T.flags[j];
            if (npos + rank < k) T.buf[npos + rank] = make_key(0.0f, (uint32_t)f);
        }
    }
    __syncthreads();
}

struct PeakOut {
    float* score;
    int64_t* ind;
    float* cls;
    float* ys;
    float* xs;
};

// standalone select: topk() outputs (utils.py:447-467)
__global__ __launch_bounds__(SEL_THREADS) void k_select_peaks(const uint64_t* __restrict__ cand, const int* __restrict__ counters,
                                                               int counter_stride, int64_t cap, int fixed_n, int k, int hw, int w,
                                                               int do_fill, PeakOut out) {
    __shared__ uint64_t buf[SORT_CAP];
    __shared__ int hist[2 * 256];
    __shared__ int misc[4];
    __shared__ int flags[SD_MAX_TOPK];
    __shared__ int alive[2];
    const int b = blockIdx.x;
    const int n = counters ? counters[b * counter_stride] : fixed_n;
    if (threadIdx.x < 2) alive[threadIdx.x] = 0;
    __syncthreads();
    __shared__ uint64_t outb[SD_MAX_TOPK];
    const Team T{(int)threadIdx.x, buf, hist, misc, flags, outb, 0, alive};
    team_select_topk<SEL_THREADS>(T, FlatSrc<SEL_THREADS, false>{cand + (int64_t)b * cap, n}, k, n, k, SORT_CAP);
    if (do_fill) fill_zero_slots<SEL_THREADS>(T, min(n, k), k);
    for (int i = threadIdx.x; i < k; i += SEL_THREADS) {
        const uint64_t key = buf[i];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw, ind = flat - cls * hw;
        const int y = ind / w, x = ind - y * w;
        out.score[(int64_t)b * k + i] = ord2f((uint32_t)(key >> 32));
        out.ind[(int64_t)b * k + i] = ind;
        out.cls[(int64_t)b * k + i] = (float)cls;
        out.ys[(int64_t)b * k + i] = (float)y;
        out.xs[(int64_t)b * k + i] = (float)x;
    }
}

// ---------------------------------------------------------------------------------------------
// Association stage, decoders.py:49-100, one block per image.  LDS arrays hold the K anchors
// and P parts; thread p scans the K anchors for part p.
// ---------------------------------------------------------------------------------------------
struct PackedLayout {
    float* anchor_out;   // (B,K,4)
    float* part_out;     // (B,P,6)
    float* part_emb;     // (B,P,2)
    float* anchor_smask; // (B,K)
    float* part_smask;   // (B,P)
    int* anchor_ind;     // (B,K)
    int* part_ind;       // (B,P)
    int* assign;         // (B,P)
    int* status;         // (B)   0 = ok, 1 = sd_decode_fused gave up waiting for a tile block (results of that image invalid)
};

__host__ __device__ inline PackedLayout packed_layout(void* packed, int B, int K, int P) {
    PackedLayout L;
    float* f = reinterpret_cast<float*>(packed);
    L.anchor_out = f;                 f += (int64_t)B * K * 4;
    L.part_out = f;                   f += (int64_t)B * P * 6;
    L.part_emb = f;                   f += (int64_t)B * P * 2;
    L.anchor_smask = f;               f += (int64_t)B * K;
    L.part_smask = f;                 f += (int64_t)B * P;
    L.anchor_ind = reinterpret_cast<int*>(f);  f += (int64_t)B * K;
    L.part_ind = reinterpret_cast<int*>(f);    f += (int64_t)B * P;
    L.assign = reinterpret_cast<int*>(f);      f += (int64_t)B * P;
    L.status = reinterpret_cast<int*>(f);
    return L;
}

struct RegMaps {
    const float* offsets;
    int64_t o_sb, o_sc;
    const float* embeddings;
    int64_t e_sb, e_sc;
};

// anchors: score/ind/cls in LDS (as_, ai_, ac_), parts likewise; writes packed outputs.
__device__ void block_group(int b, int K, int P, int w, float conf, float dist_px, const RegMaps& rm,
                            const float* as_, const int* ai_, const int* ac_,
                            const float* ps_, const int* pi_, const int* pc_,
                            float* posx, float* posy, const PackedLayout& L) {
    const int tid = threadIdx.x;
    const float* off_b = rm.offsets + (int64_t)b * rm.o_sb;
    const float* emb_b = rm.embeddings + (int64_t)b * rm.e_sb;
    // The gathers of a thread's FIRST part go out before the anchors' pass and its barrier: they do not depend on the anchors, and behind
    // the barrier they were one more dependent global round trip on the selector's critical path (same loads, same arithmetic).
    float pf_ex = 0.f, pf_ey = 0.f, pf_ox = 0.f, pf_oy = 0.f;
    if (tid < P) {
        const int ind = pi_[tid];
        pf_ex = emb_b[ind]; pf_ey = emb_b[rm.e_sc + ind]; pf_ox = off_b[ind]; pf_oy = off_b[rm.o_sc + ind];
    }
    __shared__ int n_live_s;                                    // 1 + the last anchor rank with score > conf
    if (tid == 0) n_live_s = 0;
    __syncthreads();
    int last_live = 0;
    for (int a = tid; a < K; a += (int)blockDim.x) {
        const int ind = ai_[a];
        const int y = ind / w, x = ind - y * w;
        const float score = as_[a];
        if (score > conf) last_live = a + 1;
        const float ax = (float)x + off_b[ind];                 // decoders.py:52
        const float ay = (float)y + off_b[rm.o_sc + ind];       // decoders.py:53
        const bool m = score > conf;                            // decoders.py:83
        posx[a] = m ? ax : 1e6f;                                // decoders.py:85-86
        posy[a] = m ? ay : 1e6f;
        float* ao = L.anchor_out + ((int64_t)b * K + a) * 4;
        ao[0] = ax; ao[1] = ay; ao[2] = score; ao[3] = (float)ac_[a];
        L.anchor_smask[(int64_t)b * K + a] = m ? score : -1.0f; // decoders.py:84
        L.anchor_ind[(int64_t)b * K + a] = ind;
    }
    wave_atomic_max(&n_live_s, last_live);
    lds_barrier();                                              // (not __syncthreads(): the anchors' stores stay in flight)
    const int n_live = n_live_s;
    for (int p = tid; p < P; p += (int)blockDim.x) {
        const int ind = pi_[p];
        const int y = ind / w, x = ind - y * w;
        const float score = ps_[p];
        const bool first = p == tid;
        const float ex = first ? pf_ex : emb_b[ind], ey = first ? pf_ey : emb_b[rm.e_sc + ind];    // decoders.py:66
        const float px = (float)x + (first ? pf_ox : off_b[ind]);                                   // decoders.py:67
        const float py = (float)y + (first ? pf_oy : off_b[rm.o_sc + ind]);                         // decoders.py:68
        const float ox = px + ex, oy = py + ey;                 // decoders.py:69-70
        const bool m = score > conf;                            // decoders.py:78
        const float orx = m ? ox : -1e6f, ory = m ? oy : -1e6f; // decoders.py:80-81
        // decoders.py:88-98, utils.py:433-435: d = sqrtf(fl(fl(dx*dx) + fl(dy*dy))), first minimum wins.  Anchors beyond rank n_live are all
        // masked (they arrive sorted by score, so the live ones are a prefix); a masked anchor sits at (1e6, 1e6) (decoders.py:85-86) and a
        // masked part starts from (-1e6, -1e6) (:80-81): neither comes within dist_px (<= the map side) of anything -- masked parts are
        // unassigned without a scan, live parts scan the ranks below n_live only.  sqrtf is monotone: a sum that is not smaller than the best
        // sum cannot give a smaller distance, so the square root is taken only for the few candidates that lower the running sum and the
        // reference's strict `<` on the rounded distances decides there (equal rounded distances: the earlier anchor stays).
        float best = INFINITY, best_s = INFINITY;
        int best_a = 0;
        if (m && dist_px < 1e5f) {
            for (int a = 0; a < n_live; ++a) {
                const float dx = orx - posx[a], dy = ory - posy[a];
                const float sx = dx * dx, sy = dy * dy;
                const float ss = sx + sy;
                if (ss < best_s) {
                    const float d = sqrtf(ss);
                    best_s = ss;
                    if (d < best) { best = d; best_a = a; }     // strict <: lowest anchor rank wins ties
                }
            }
        } else {
            for (int a = 0; a < K; ++a) {                       // (absurd thresholds: the reference's full scan)
                const float dx = orx - posx[a], dy = ory - posy[a];
                const float sx = dx * dx, sy = dy * dy;
                const float d = sqrtf(sx + sy);
                if (d < best) { best = d; best_a = a; }
            }
        }
        float* po = L.part_out + ((int64_t)b * P + p) * 6;
        po[0] = px; po[1] = py; po[2] = score; po[3] = (float)pc_[p]; po[4] = ox; po[5] = oy;
        L.part_emb[((int64_t)b * P + p) * 2 + 0] = ex;
        L.part_emb[((int64_t)b * P + p) * 2 + 1] = ey;
        L.part_smask[(int64_t)b * P + p] = m ? score : -1.0f;   // decoders.py:79
        L.part_ind[(int64_t)b * P + p] = ind;
        L.assign[(int64_t)b * P + p] = (best < dist_px) ? best_a : -1;   // decoders.py:100
    }
    if (tid == 0) L.status[b] = 0;
}

// fused: select anchors and parts side by side (two teams), then associate (2nd and last launch of sd_decode)
__global__ __launch_bounds__(2 * SEL_THREADS) void k_select_group(const uint64_t* __restrict__ cand0, const uint64_t* __restrict__ cand1,
                                                                   const int* __restrict__ counters, int M, int N, int h, int w,
                                                                   int K, int P, float conf, float dist_px, RegMaps rm,
                                                                   void* packed, int B) {
    __shared__ uint64_t buf[2][SORT_CAP];
    __shared__ int hist[2][2 * 256];
    __shared__ int misc[2][4];
    __shared__ int flags[2][SD_MAX_TOPK];
    __shared__ int alive[2];
    __shared__ float as_[SD_MAX_TOPK], ps_[SD_MAX_TOPK], posx[SD_MAX_TOPK], posy[SD_MAX_TOPK];
    __shared__ int ai_[SD_MAX_TOPK], ac_[SD_MAX_TOPK], pi_[SD_MAX_TOPK], pc_[SD_MAX_TOPK];
    const int b = blockIdx.x, hw = h * w;
    const int team = threadIdx.x >> 9, tid = threadIdx.x & (SEL_THREADS - 1);
    __shared__ uint64_t outb[2][SD_MAX_TOPK];
    const Team T{tid, buf[team], hist[team], misc[team], flags[team], outb[team], team, alive};

    const int n0 = counters[(b * 2 + 0) * CNT_STRIDE], n1 = counters[(b * 2 + 1) * CNT_STRIDE];
    // identical barrier sequence for both teams: the path comes from the longer list and the larger k
    const int n = team ? n1 : n0, k = team ? P : K;
    const uint64_t* cand = team ? cand1 + (int64_t)b * N * hw : cand0 + (int64_t)b * M * hw;
    team_select_topk<SEL_THREADS>(T, FlatSrc<SEL_THREADS, false>{cand, n}, k, max(n0, n1), max(K, P), SORT_CAP);
    fill_zero_slots<SEL_THREADS>(T, min(n, k), k);
    float* os = team ? ps_ : as_;
    int* oi = team ? pi_ : ai_;
    int* oc = team ? pc_ : ac_;
    for (int i = tid; i < k; i += SEL_THREADS) {
        const uint64_t key = T.buf[i];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw;
        os[i] = ord2f((uint32_t)(key >> 32)); oi[i] = flat - cls * hw; oc[i] = cls;
    }
    __syncthreads();
    const PackedLayout L = packed_layout(packed, B, K, P);
    block_group(b, K, P, w, conf, dist_px, rm, as_, ai_, ac_, ps_, pi_, pc_, posx, posy, L);
}


// ---------------------------------------------------------------------------------------------
// MAP-PARALLEL decoder (sd_decode on large geometries: BASELINE configs[4], 1024x1024 inputs with 8 + 8 maps = 16 k tiles per batch
// of 16).  What the two-launch pair above costs there (rocprofv3, profiles/r03_decode_variants.txt): k_nms_tile<1> 58 us -- its
// 16 384 blocks append to 32 per-(image, group) counters, 512 same-address atomics each, serialised at the memory side -- and
// k_select_group 55 us: ONE block per image radix-selects 11 k + 7 k candidates from global memory, ~10 us per pass of dependent
// round trips, on 16 of 256 CUs.  Here:
//   1. k_nms_slots<TH>: the tile pass without any global atomic: a tile owns TW x TH candidate slots and one count word (plain stores);
//   2. k_select_map: one block per MAP (B x (M + N) blocks): the reference's own first stage -- per-class top-k (utils.py:451) -- on
//      the map's tiles: counts -> prefix -> keys into LDS -> radix select; writes <= k keys per map;
//   3. k_rank_maps: the reference's second stage (utils.py:459: top-k over the C x k stage-1 candidates) WITHOUT a second selection:
//      the stage-1 lists are sorted, a key's final rank is a sum of binary searches -- one block per map again;
//   4. k_group_wide: zero slots + association, 64 parts per block.
// (A first version merged and associated in ONE block per image: 41 us for 16 images -- in-kernel timestamps: 22 us of radix select +
// bitonic sort of 512 keys, 18-25 us for the K x P distance scan on one CU.)
// Two-stage top-k == global top-k as a set (any global top-k element is in its class's top-k), and the final order is the same
// total order of the same keys: results are bit-identical to k_select_group's.
// ---------------------------------------------------------------------------------------------
template <int TH_>
__global__ __launch_bounds__(256) void k_nms_slots(Group g0, Group g1, int h, int w, int tiles_x, int tiles, float min_score,
                                                    uint64_t* __restrict__ cand, int* __restrict__ tile_cnt) {
    constexpr int LH_ = TH_ + 2 * HALO, CAP_ = TW * TH_;
    __shared__ float S[LH_][LW];
    __shared__ float Hm[LH_][TW];
    __shared__ int keep_n;
    const int tid = threadIdx.x;
    const int C = g0.C + g1.C;
    const int64_t blk = blockIdx.x;                            // ((b * C) + m) * tiles + tile
    const int tile = (int)(blk % tiles);
    const int bm = (int)(blk / tiles);
    const int b = bm / C, m = bm - b * C;
    const int grp = (m >= g0.C) ? 1 : 0;
    const Group g = grp ? g1 : g0;
    const int c = grp ? m - g0.C : m;
    const int tx0 = (tile % tiles_x) * TW;
    const int ty0 = (tile / tiles_x) * TH_;
    const float* plane = g.p + (int64_t)b * g.sb + (int64_t)c * g.sc;
    uint64_t* mine = cand + blk * CAP_;
    if (tid == 0) keep_n = 0;
    constexpr int NLD = (LH_ * LW + 255) / 256;
    float ld[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + j * 256;
        const int r = i / LW, cc = i - r * LW;
        const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
        const bool ok = i < LH_ * LW && y >= 0 && y < h && x >= 0 && x < w;
        ld[j] = plane[ok ? (int64_t)y * w + x : 0];
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + j * 256;
        const int r = i / LW, cc = i - r * LW;
        const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
        const bool ok = y >= 0 && y < h && x >= 0 && x < w;
        if (i < LH_ * LW) S[r][cc] = ok ? clamped_sigmoid(ld[j]) : -INFINITY;
    }
    __syncthreads();
    for (int i = tid; i < LH_ * TW; i += 256) {
        const int r = i / TW, cc = i - r * TW;
        float mx = fmaxf(fmaxf(S[r][cc], S[r][cc + 1]), fmaxf(S[r][cc + 2], S[r][cc + 3]));
        Hm[r][cc] = fmaxf(mx, S[r][cc + 4]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (TW * TH_) / 256; ++j) {
        const int i = tid + j * 256;
        const int r = i / TW, cc = i - r * TW;
        const int y = ty0 + r, x = tx0 + cc;
        float mx = fmaxf(fmaxf(Hm[r][cc], Hm[r + 1][cc]), fmaxf(Hm[r + 2][cc], Hm[r + 3][cc]));
        mx = fmaxf(mx, Hm[r + 4][cc]);
        const float v = S[r + HALO][cc + HALO];
        if ((y < h) && (x < w) && (v == mx) && (v >= min_score)) {            // `>=`: see k_nms_tile
            const int slot = atomicAdd(&keep_n, 1);                            // LDS
            mine[slot] = make_key(v, (uint32_t)(c * h * w + y * w + x));
        }
    }
    __syncthreads();
    if (tid == 0) tile_cnt[blk] = keep_n;
}

// ---------------------------------------------------------------------------------------------
// The tile pass in the LOGIT domain (k_nms_slots_v): k_nms_slots is VALU-bound -- ~80 instructions per pixel, a third of them the exact
// sigmoid (expf + IEEE division) of every staged pixel, the rest scalar index arithmetic -- at 48 us for 67 MB (1.4 TB/s).  The clamped
// sigmoid is monotone non-decreasing in fp32 (checked over ALL 2^32 bit patterns on the device: sd_selfcheck_sigmoid), so
//     max over the window of sigma(x_j) == sigma(max over the window of x_j)        and
//     pixel c survives  <=>  sigma(x_c) == sigma(m),   m = the window's maximum logit:
//   * x_c == m: survives, no sigmoid needed for the decision;
//   * x_c <  m - margin(m): sigma(x_c) < sigma(m) strictly (the margin table is part of the same exhaustive check): suppressed;
//   * in between (a near-tie, or a saturated window: sigma clamps at both ends, whole plateaus survive): the two sigmoids decide.
// Sigmoids are evaluated only for the compacted survivors (~4 % of the pixels), whose score the key needs anyway.  Staging, row maxima
// and column maxima work on float4 (16-byte global loads, ds_read / ds_write_b128): one index computation per four pixels.
// Needs w % 4 == 0 and 16-byte aligned planes (map_view gives them); k_nms_slots stays as the general kernel.  Same survivors, same
// keys as k_nms_slots.
// ---------------------------------------------------------------------------------------------
// margin(m): INFINITY where the clamp may be active (|m| >= 13: sigma(+-13.8) are the clamp values), 1e-4 below m = 4 (there one ulp of
// sigma is < 2e-5 of a logit step), 1.0 up to 13 (sigma(13) - sigma(12) = 65 ulp).  Three instructions; the table is deliberately
// coarse: a wider margin only sends a few more pixels around strong peaks through the exact compare.
__device__ __forceinline__ float nms_margin(float m) {
    return fabsf(m) >= 13.0f ? INFINITY : (m < 4.0f ? 1e-4f : 1.0f);
}
__device__ __forceinline__ float max5(float a, float b, float c, float d, float e) { return fmaxf(fmaxf(fmaxf(a, b), c), fmaxf(d, e)); }

template <int TH_>
__global__ __launch_bounds__(256) void k_nms_slots_v(Group g0, Group g1, int h, int w, int tiles_x, int tiles, float min_score,
                                                      uint64_t* __restrict__ cand, int* __restrict__ tile_cnt) {
    constexpr int LH_ = TH_ + 2 * HALO, CAP_ = TW * TH_;
    constexpr int SV = (TW + 8) / 4, OV = TW / 4;              // float4 per staged row (columns tx0 - 4 .. tx0 + TW + 3) / per output row
    __shared__ float4 S[LH_][SV];
    __shared__ float4 Hm[LH_][OV];
    __shared__ unsigned short list[CAP_];                       // survivors / near-ties: local pixel index | near << 15
    __shared__ int keep_n, out_n;
    const int tid = threadIdx.x;
    const int C = g0.C + g1.C;
    const int64_t blk = blockIdx.x;                            // ((b * C) + m) * tiles + tile
    const int tile = (int)(blk % tiles);
    const int bm = (int)(blk / tiles);
    const int b = bm / C, m = bm - b * C;
    const int grp = (m >= g0.C) ? 1 : 0;
    const Group g = grp ? g1 : g0;
    const int c = grp ? m - g0.C : m;
    const int tx0 = (tile % tiles_x) * TW;
    const int ty0 = (tile / tiles_x) * TH_;
    const float* plane = g.p + (int64_t)b * g.sb + (int64_t)c * g.sc;
    uint64_t* mine = cand + blk * CAP_;
    if (tid == 0) { keep_n = 0; out_n = 0; }
    constexpr int NLD = (LH_ * SV + 255) / 256;
    float4 ld[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int u = tid + j * 256;
        const int r = u / SV, q = u - r * SV;
        const int y = ty0 + r - HALO, x0 = tx0 - 4 + 4 * q;
        const bool ok = u < LH_ * SV && y >= 0 && y < h && x0 >= 0 && x0 < w;
        ld[j] = *reinterpret_cast<const float4*>(plane + (ok ? (int64_t)y * w + x0 : 0));
    }
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int u = tid + j * 256;
        const int r = u / SV, q = u - r * SV;
        const int y = ty0 + r - HALO, x0 = tx0 - 4 + 4 * q;
        const bool ok = y >= 0 && y < h && x0 >= 0 && x0 < w;
        if (u < LH_ * SV) S[r][q] = ok ? ld[j] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    __syncthreads();
    for (int v = tid; v < LH_ * OV; v += 256) {                 // horizontal 5-maxima of output columns 4q .. 4q + 3
        const int r = v / OV, q = v - r * OV;
        const float4 a = S[r][q], bq = S[r][q + 1], cq = S[r][q + 2];       // staged columns 4q .. 4q + 11 = output columns 4q - 4 .. 4q + 7
        Hm[r][q] = make_float4(max5(a.z, a.w, bq.x, bq.y, bq.z), max5(a.w, bq.x, bq.y, bq.z, bq.w), max5(bq.x, bq.y, bq.z, bq.w, cq.x),
                               max5(bq.y, bq.z, bq.w, cq.x, cq.y));
    }
    __syncthreads();
    for (int v = tid; v < TH_ * OV; v += 256) {
        const int r = v / OV, q = v - r * OV;
        const float4 h0 = Hm[r][q], h1 = Hm[r + 1][q], h2 = Hm[r + 2][q], h3 = Hm[r + 3][q], h4 = Hm[r + 4][q];
        const float4 xc = S[r + HALO][q + 1];
        const float mx[4] = {max5(h0.x, h1.x, h2.x, h3.x, h4.x), max5(h0.y, h1.y, h2.y, h3.y, h4.y), max5(h0.z, h1.z, h2.z, h3.z, h4.z),
                             max5(h0.w, h1.w, h2.w, h3.w, h4.w)};
        const float xv[4] = {xc.x, xc.y, xc.z, xc.w};
        const bool row_in = ty0 + r < h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool top = xv[e] == mx[e];
            const bool near = !(mx[e] - xv[e] > nms_margin(mx[e]));             // (also true for top; NaN-safe: inf - inf fails `>`)
            if (row_in && tx0 + 4 * q + e < w && near) {
                const int slot = atomicAdd(&keep_n, 1);                        // LDS
                list[slot] = (unsigned short)((r * TW + 4 * q + e) | (top ? 0 : 0x8000));
            }
        }
    }
    __syncthreads();
    const int n = keep_n;
    const float* Sf = reinterpret_cast<const float*>(&S[0][0]);
    const float* Hf = reinterpret_cast<const float*>(&Hm[0][0]);
    for (int i = tid; i < n; i += 256) {
        const int ent = list[i];
        const int li = ent & 0x7fff, r = li / TW, cx = li - r * TW;
        const float x = Sf[(r + HALO) * (SV * 4) + cx + 4];
        const float v = clamped_sigmoid(x);
        bool ok = v >= min_score;                                              // `>=`: see k_nms_tile
        if (ok && (ent & 0x8000)) {                                            // near-tie / saturated window: the reference's own compare
            const float mxl = max5(Hf[r * TW + cx], Hf[(r + 1) * TW + cx], Hf[(r + 2) * TW + cx], Hf[(r + 3) * TW + cx], Hf[(r + 4) * TW + cx]);
            ok = clamped_sigmoid(mxl) == v;
        }
        if (ok) {
            const int slot = atomicAdd(&out_n, 1);
            mine[slot] = make_key(v, (uint32_t)(c * h * w + (ty0 + r) * w + tx0 + cx));
        }
    }
    __syncthreads();
    if (tid == 0) tile_cnt[blk] = out_n;
}

// Exhaustive check of the two properties k_nms_slots_v rests on, over every fp32 bit pattern in numeric order o -> ord2f(o):
//   out[0] += 1 for every consecutive pair with clamped_sigmoid(next) < clamped_sigmoid(this)             (monotonicity)
//   out[1] += 1 for every m with a finite margin where the largest x < m - margin(m) has sigma(x) >= sigma(m) (margin table)
//   out[2] += number of values visited
__global__ __launch_bounds__(256) void k_selfcheck_sigmoid(unsigned long long* out) {
    unsigned long long bad_mono = 0, bad_margin = 0, seen = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t o = (uint64_t)blockIdx.x * 256 + threadIdx.x; o < (1ull << 32); o += stride) {
        const float x = ord2f((uint32_t)o);
        if (x != x) continue;
        ++seen;
        const float sx = clamped_sigmoid(x);
        if (o + 1 < (1ull << 32)) {
            const float y = ord2f((uint32_t)(o + 1));
            if (y == y && clamped_sigmoid(y) < sx) ++bad_mono;
        }
        const float mg = nms_margin(x);
        if (mg != INFINITY) {
            const float t = x - mg;
            const float below = ord2f(f2ord(t) - 1u);
            if (!(clamped_sigmoid(below) < sx)) ++bad_margin;
        }
    }
    if (bad_mono) atomicAdd(&out[0], bad_mono);
    if (bad_margin) atomicAdd(&out[1], bad_margin);
    atomicAdd(&out[2], seen);
}

// ---------------------------------------------------------------------------------------------
// k_map_stream_select: tile pass AND the per-map selection in one kernel, one 512-thread block per (image, map).  At BASELINE
// configs[4] a batch is 16 x 16 = 256 maps of 256 KB: one map per CU, each CU streaming at its share of the HBM rate (~20 GB/s), no
// candidate list in global memory, no tile counts, no second launch for the first selection stage.
//   * streaming: a wave owns a strip of up to 256 columns (one float4 per lane) and walks 16 output rows (+ 4 halo rows); the horizontal
//     neighbours come from the adjacent lanes (cross-lane moves; the two edge lanes load their halo), the vertical window is five rows of
//     horizontal maxima in registers: no LDS, no barrier, no index arithmetic per pixel; rows are requested five ahead of their use;
//   * the logit-domain survivor rule of k_nms_slots_v; candidates are appended (one LDS atomic per wave and row) as raw entries
//     {logit, near flag, pixel} to a 4096-entry LDS stage, and the sigmoids are taken afterwards over the compacted entries;
//   * selection from LDS (radix select, ranking / register bitonic sort) and one store of the map's sorted top-k, zero padded to k:
//     the later stages need no counts.
// A map with more than 4096 candidates (plateaus; > 6 % of all pixels) is walked a second time with the keys going to the global
// candidate list and selected from there.  Needs w % 4 == 0 and 16-byte aligned planes.
// ---------------------------------------------------------------------------------------------
// entries of a WAVE's segment of the stage: 4096 entries over the waves of a block of 8 or 16 waves; 256 per wave in smaller blocks
// (a wave of those walks one or two bands of a split map: 512 entries = 9 % of two 11-row bands of 256 columns)
__host__ __device__ constexpr int stream_seg(int waves) { return waves >= 8 ? 4096 / waves : 512; }
#ifndef SD_STREAM_DEPTH_SMALL
#define SD_STREAM_DEPTH_SMALL 0          // (timing experiments: 5 = the round-4 ring everywhere)
#endif
// rows a wave requests ahead: in blocks of up to 8 waves every row of an 11-row band (15 rows, one round trip: 6 registers per row and lane)
// and ten of the twenty rows of a 16-row band; the five of the window in 16-wave blocks (128 registers per lane there: ten rows ahead spill,
// measured on the 256 x 256 maps of the stress shape 36.5 -> 41.8 us)
__host__ __device__ constexpr int stream_depth(int nt, int nr) {
    return SD_STREAM_DEPTH_SMALL ? SD_STREAM_DEPTH_SMALL : (nt <= 512 ? (nr % 10 == 0 ? 10 : nr) : (nr % 5 == 0 ? 5 : nr));
}
// Round 5: a map may be SPLIT over `splits` blocks (blockIdx.x = (image * maps + map) * splits + part): a block walks the row bands
// [part * bands / splits, (part + 1) * bands / splits) of its map (the halo rows of a band are read from the map, whoever owns them) and
// stores ITS sorted top-k as one more stage-1 list -- k_rank_maps / k_rank_group merge the lists of a group by rank whatever map or
// part of a map they come from (the class travels in the key).  At BASELINE configs[2] (bs = 64, 3 maps of 128 x 128) one block per map
// was 192 blocks of 12 busy waves on 256 CUs; three parts per map are 576 blocks of 4 waves, every wave with one 11-row band.

// lane i <- lane i - 1 / lane i + 1 of the wave (DPP wave_shr:1 / wave_shl:1: one VALU move, no trip through the LDS crossbar; bound_ctrl:
// the lane without a source reads 0 and the destination needs no initialising move -- the callers replace that lane's value anyway)
__device__ __forceinline__ float lane_from_left(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_from_right(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, true));
}

// max of three in ONE instruction (fmaxf chains compile to v_max_f32 pairs plus NaN canonicalisation moves: 47 instructions for the
// 14 three-way maxima of a row)
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float min3f(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// sixteen bytes of -inf: what a lane outside the map, a row outside the map and the 62 inner lanes' halo read (the pointer is chosen when
// the row is REQUESTED: no masking of the loaded values afterwards -- six selects per row and lane in the two-bands-per-wave form)
__device__ __attribute__((aligned(16))) const float g_neg_inf4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};

// Candidates of a wave go to the wave's OWN segment of the stage (STREAM_SEG entries), the fill level lives in a scalar register: one
// ballot per row ("this lane has a candidate") ranks the lanes, no atomic -- per-lane LDS atomics on one block-wide counter serialise
// (~20 cycles each, 80 per row over the block's waves: the append then cost as much as the whole rest of the walk), per-element ballots
// cost ~100 instructions per row.  A lane with a SECOND candidate among its four pixels (ties only: 5x5 maxima are >= 3 apart) puts it
// on the small block-wide `extra` list through an LDS atomic.  count[0] = largest fill level of a segment seen (block-wide maximum:
// the overflow test), count[1] = entries on the extra list; the wave's fill level goes to seg_fill[wave].
constexpr int STREAM_EXTRA = 256;
// HALF (round 5; maps up to 128 columns wide): a strip of 128 columns is 32 lanes of one float4, so the two halves of a wave walk TWO bands
// side by side (lanes 0-31: band u, lanes 32-63: band u + 1) -- at the cfg shape half of every wave used to idle beyond the map's right
// edge.  Half as many waves to start (the start-up spread of a launch is ~2 ns per wave), half the vector instructions per pixel.
template <bool INLINE_KEYS, int NT, int ROWS, bool HALF>
__device__ __forceinline__ void stream_map(const float* __restrict__ plane, int h, int w, int c, float min_score, float min_logit, uint64_t* stage,
                                           float* mxs, int* count, uint64_t* __restrict__ gkeys, int* seg_fill, uint64_t* extra, float* extra_mx,
                                           int band_lo, int band_hi) {
    constexpr int STREAM_WAVES = NT / 64, STREAM_SEG = stream_seg(STREAM_WAVES);
    // WIN = the 5-row window of horizontal maxima; DEPTH = rows requested ahead (round 5, late: the two used to be ONE ring of five, and a
    // wave's 15 / 20 rows were three / four dependent round trips to memory -- `profiles/r05_decode_trace_cfg.txt`: 7-10 us of walking for
    // 15 rows whose arithmetic is ~0.1 us each.  Now a wave of the small blocks asks for ALL its rows at once, one round trip)
    constexpr int R = ROWS, NR = R + 4, WIN = 5, DEPTH = stream_depth(NT, NR);
    static_assert(NR % DEPTH == 0 && (DEPTH == NR || DEPTH % WIN == 0), "the row loop is unrolled by the request depth: all rows, or a multiple of the window");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int strips = (w + 255) >> 8;
    const float NEG = -INFINITY;
    int fill = 0;                                               // (wave-uniform) entries in this wave's segment
    uint64_t* seg = stage + wave * STREAM_SEG;
    float* seg_mx = mxs + wave * STREAM_SEG;
    constexpr int PER = HALF ? 2 : 1;                           // units of work a wave walks at once
    const int sl = HALF ? (lane & 31) : lane;                   // lane within its strip
    const bool edge_l = sl == 0, edge_r = sl == (HALF ? 31 : 63);
    const int units = strips * (band_hi - band_lo);
    const bool has_cut = min_logit > -INFINITY;                 // (uniform) annotations-only mode
    for (int unit0 = wave * PER; unit0 < units; unit0 += STREAM_WAVES * PER) {
        const int unit = unit0 + (HALF ? (lane >> 5) : 0);      // (HALF: strips == 1, the unit is the band)
        const bool unit_in = !HALF || unit < units;
        const int sy = band_lo + (HALF ? unit : unit / strips), sx = HALF ? 0 : unit % strips;
        const int x0 = sx * 256 + sl * 4;
        const bool col_in = x0 < w && unit_in;
        const bool narrow = HALF || sx * 256 + 256 > w;         // (wave-uniform) the strip has lanes beyond the map's right edge / an empty half
        const int y0 = sy * R;                                  // (HALF: differs between the two halves of the wave)
        // the edge lanes' halo: the first lane of a strip needs the two columns left of it, the last the two right of it (8-byte aligned pairs)
        const int hx = edge_l ? x0 - 2 : x0 + 4;
        const bool halo_in = (edge_l || edge_r) && hx >= 0 && hx < w && unit_in;
        // row pointers advance by w per request; lanes outside the map (and the 62 inner lanes' halo) read g_neg_inf4
        const float* const negp = g_neg_inf4;
        const float* pv = col_in ? plane + x0 : negp;
        const float* ph = halo_in ? plane + hx : negp;
        const int64_t vstep = col_in ? w : 0, hstep = halo_in ? w : 0;
        float4 v[DEPTH], hm[WIN];
        float2 hl[DEPTH];
        float4 c1 = make_float4(NEG, NEG, NEG, NEG), c2 = c1;   // the rows one and two above the row being staged
        auto request = [&](int j, int slot) {
            const int y = y0 + j - 2;
            const bool rin = (unsigned)y < (unsigned)h;         // rows outside the map read -inf (wave-uniform unless HALF)
            v[slot] = *reinterpret_cast<const float4*>(rin ? pv + y * vstep : negp);
            hl[slot] = *reinterpret_cast<const float2*>(rin ? ph + y * hstep : negp);
        };
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) request(u, u);
        for (int jb = 0; jb < NR; jb += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int j = jb + u;
                const float4 cur = v[u];
                const float2 hh = hl[u];
                if (j + DEPTH < NR) request(j + DEPTH, u);             // the slot is free again: row j + DEPTH goes out
                float lz = lane_from_left(cur.z), lw = lane_from_left(cur.w), rx = lane_from_right(cur.x), ry = lane_from_right(cur.y);
                if (edge_l) { lz = hh.x; lw = hh.y; }                  // (HALF: lane 32's left neighbour is lane 31 of the OTHER band: replaced like lane 0's)
                if (edge_r) { rx = hh.x; ry = hh.y; }
                const float ma = max3f(cur.x, cur.y, cur.z), mb = max3f(cur.y, cur.z, cur.w);
                hm[u % WIN] = make_float4(max3f(ma, lz, lw), max3f(ma, lw, cur.w), max3f(mb, cur.x, rx), max3f(mb, rx, ry));
                const float4 centre = c2;                               // row j - 2
                c2 = c1; c1 = cur;
                const int yo = y0 + j - 4;                              // output row: window rows j - 4 .. j (all WIN slots), centre j - 2
                if (j < 4) continue;
                if constexpr (!HALF) { if (yo >= h) continue; }         // (wave-uniform)
                // HALF: an output row beyond the map in ONE half of the wave: no candidates there (a window entirely outside the map is all
                // -inf, and inf - inf fails the `>` of the survivor rule: it must be masked, not left to the arithmetic)
                const bool row_ok = !HALF || yo < h;
#if defined(SD_STREAM_ABL) && SD_STREAM_ABL == 2                         // timing experiment: loads + horizontal maxima only
                if (hm[u % WIN].x != 12345.f) continue;
#endif
                const float xv[4] = {centre.x, centre.y, centre.z, centre.w};
                const float xhi = max3f(max3f(xv[0], xv[1], xv[2]), xv[3], xv[3]);
                // with a score threshold most rows have no pixel above its logit anywhere in the wave: they skip the column maxima and the
                // survivor test altogether (pixels outside the map are -inf; the dense scenes' walk was bound by vector instruction issue:
                // 16 waves per CU x 20 rows x ~90 instructions)
                if (has_cut && __ballot(xhi >= min_logit) == 0ull) continue;
                const float mx[4] = {max3f(max3f(hm[0].x, hm[1].x, hm[2].x), hm[3].x, hm[4].x), max3f(max3f(hm[0].y, hm[1].y, hm[2].y), hm[3].y, hm[4].y),
                                     max3f(max3f(hm[0].z, hm[1].z, hm[2].z), hm[3].z, hm[4].z), max3f(max3f(hm[0].w, hm[1].w, hm[2].w), hm[3].w, hm[4].w)};
                const uint32_t pix = (uint32_t)(yo * w + x0);
                if (INLINE_KEYS) {                                       // (second walk of an overflowing map: keys with their sigmoids at once)
                    int cnt = 0;
                    bool cand[4];
                    float sc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        cand[e] = !(mx[e] - xv[e] > nms_margin(mx[e])) && !(narrow && !col_in) && row_ok && xv[e] >= min_logit;
                        sc[e] = 0.f;
                        if (cand[e]) {
                            sc[e] = clamped_sigmoid(xv[e]);
                            cand[e] = sc[e] >= min_score && (xv[e] == mx[e] || clamped_sigmoid(mx[e]) == sc[e]);
                        }
                        cnt += cand[e] ? 1 : 0;
                    }
                    if (cnt == 0) continue;
                    int slot = atomicAdd(count, cnt);                   // LDS
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cand[e]) gkeys[slot++] = make_key(sc[e], (uint32_t)(c * h * w) + pix + e);
                    continue;
                }
                const bool live = !(narrow && !col_in) && row_ok;
                {   // Quiet rows leave here: with every window maximum of the lane's four pixels in (-13, 4) the margin is 1e-4 for all four, so
                    // "no pixel within 1e-4 of its maximum" is the survivor rule's own answer, in 13 instructions instead of 24 + the flag logic
                    // (a lane beyond the map holds -inf maxima: not quiet, dropped by `live` as below)
                    const float dmin = min3f(min3f(mx[0] - xv[0], mx[1] - xv[1], mx[2] - xv[2]), mx[3] - xv[3], mx[3] - xv[3]);
                    const float mhi = max3f(max3f(mx[0], mx[1], mx[2]), mx[3], mx[3]), mlo = min3f(min3f(mx[0], mx[1], mx[2]), mx[3], mx[3]);
                    // ... and a lane whose four logits are all below min_logit has nothing the score threshold would keep (annotations-only
                    // mode: without this every local maximum of the background -- one per ~13 pixels, some in every row -- went through
                    // the append below to be dropped by the sigmoid pass: 3.3 of the 5.9 us a wave walked at the cfg shape)
                    const bool quiet = (dmin > 1e-4f && mhi < 4.0f && mlo > -13.0f) || xhi < min_logit;
#if !defined(SD_STREAM_ABL) || SD_STREAM_ABL != 1
                    if (__ballot(!quiet && live) == 0ull) continue;    // (wave-uniform)
#endif
                }
                // (lanes beyond the map hold -inf: inf - inf fails `>`, they are dropped by the `narrow` term)
                const bool c0r = !(mx[0] - xv[0] > nms_margin(mx[0])), c1r = !(mx[1] - xv[1] > nms_margin(mx[1]));
                const bool c2r = !(mx[2] - xv[2] > nms_margin(mx[2])), c3r = !(mx[3] - xv[3] > nms_margin(mx[3]));
                const bool c0 = c0r && live && xv[0] >= min_logit, c1e = c1r && live && xv[1] >= min_logit;
                const bool c2e = c2r && live && xv[2] >= min_logit, c3 = c3r && live && xv[3] >= min_logit;
                const bool any = c0 || c1e || c2e || c3;
#if defined(SD_STREAM_ABL) && SD_STREAM_ABL == 1                         // timing experiment: no append (results wrong by design)
                if (mx[0] != 12345.f) continue;
#endif
                const unsigned long long has = __ballot(any);
                if (has == 0ull) continue;                              // (wave-uniform)
                if (any) {
                    // the lane's FIRST candidate goes to the wave's segment at fill + (lanes below with a candidate)
                    const int e0 = c0 ? 0 : (c1e ? 1 : (c2e ? 2 : 3));
                    const float x0v = c0 ? xv[0] : (c1e ? xv[1] : (c2e ? xv[2] : xv[3]));
                    const float m0v = c0 ? mx[0] : (c1e ? mx[1] : (c2e ? mx[2] : mx[3]));
                    const int slot = fill + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u));
                    if (slot < STREAM_SEG) {
                        const bool top = x0v == m0v;
                        seg[slot] = ((uint64_t)__float_as_uint(x0v) << 32) | (top ? 0u : 0x80000000u) | (pix + e0);
                        if (!top) seg_mx[slot] = m0v;
                    }
                    const int more = (c0 ? 1 : 0) + (c1e ? 1 : 0) + (c2e ? 1 : 0) + (c3 ? 1 : 0) - 1;
                    if (more > 0) {                                     // further candidates of the same four pixels: ties / near-ties only
                        int xs = atomicAdd(&count[1], more);            // LDS
                        const bool ce[4] = {c0, c1e, c2e, c3};
#pragma unroll
                        for (int e = 1; e < 4; ++e) {
                            if (!ce[e] || e == e0) continue;
                            if (xs < STREAM_EXTRA) {
                                extra[xs] = ((uint64_t)__float_as_uint(xv[e]) << 32) | (xv[e] == mx[e] ? 0u : 0x80000000u) | (pix + e);
                                extra_mx[xs] = mx[e];
                            }
                            ++xs;
                        }
                    }
                }
                fill += __popcll(has);
            }
        }
    }
    if (!INLINE_KEYS && lane == 0) {
        seg_fill[wave] = min(fill, STREAM_SEG);
        atomicMax(&count[0], fill);
    }
}

template <int STREAM_THREADS, int ROWS, bool HALF = false>
__global__ __launch_bounds__(STREAM_THREADS) void k_map_stream_select(Group g0, Group g1, int h, int w, float min_score, float min_logit, int K, int P,
                                                                       uint64_t* __restrict__ cand, uint64_t* __restrict__ stage1, int splits) {
    constexpr int STREAM_WAVES = STREAM_THREADS / 64, STREAM_SEG = stream_seg(STREAM_WAVES), STREAM_CAP = STREAM_WAVES * STREAM_SEG;
    constexpr int OUT_CAP = STREAM_CAP > SD_MAX_TOPK ? STREAM_CAP : SD_MAX_TOPK;    // `stage` is also the selection's output scratch (np2k keys)
    __shared__ uint64_t stage[OUT_CAP];                                 // raw entries, then scratch of the selection (T.out)
    __shared__ uint64_t buf[OUT_CAP + STREAM_EXTRA];                    // keys (segments + extra list)
    __shared__ float mxs[STREAM_CAP];
    __shared__ int hist[2 * 256];
    __shared__ int misc[4];
    __shared__ int alive[2];
    __shared__ int counts[3];                                           // [0] fullest segment / global keys, [1] extra entries, [2] keys
    __shared__ int seg_fill[STREAM_WAVES];
    __shared__ uint64_t extra[STREAM_EXTRA];
    __shared__ float extra_mx[STREAM_EXTRA];
    const int tid = threadIdx.x;
    const int C = g0.C + g1.C;
    const int part = blockIdx.x % splits, bm = blockIdx.x / splits, b = bm / C, m = bm - b * C;
    const int bands = (h + ROWS - 1) / ROWS;
    const int band_lo = part * bands / splits, band_hi = (part + 1) * bands / splits;
    const int grp = (m >= g0.C) ? 1 : 0;
    const Group g = grp ? g1 : g0;
    const int c = grp ? m - g0.C : m;
    const int k = grp ? P : K, kmax = max(K, P);
    const int hw = h * w;
    const float* plane = g.p + (int64_t)b * g.sb + (int64_t)c * g.sc;
    if (tid < 3) counts[tid] = 0;
    if (tid < 2) alive[tid] = 0;
    [[maybe_unused]] const int trace0 = blockIdx.x == 0 ? 6400 : ((int)blockIdx.x == g0.C * splits ? 6420 : (blockIdx.x == gridDim.x - 1 ? 6440 : -100));
    SD_TRACE(trace0 + 0);
    SD_TRACE(blockIdx.x < 592 ? 7000 + (int)blockIdx.x : -1);          // (trace builds: start / end of every block of the first 592)
    __syncthreads();
    stream_map<false, STREAM_THREADS, ROWS, HALF>(plane, h, w, c, min_score, min_logit, stage, mxs, counts, nullptr, seg_fill, extra, extra_mx, band_lo, band_hi);
    SD_TRACE(trace0 + 1);
    // sigmoids of the compacted entries only: every wave converts its own segment as soon as it has walked its rows
    auto convert = [&](uint64_t ent, float mxv) {
        const float x = __uint_as_float((uint32_t)(ent >> 32));
        const float v = clamped_sigmoid(x);
        bool ok = v >= min_score;                                       // `>=`: see k_nms_tile
        if (ok && ((uint32_t)ent & 0x80000000u)) ok = clamped_sigmoid(mxv) == v;       // near-tie / saturated window
        const int slot = alloc_slot(&counts[2], ok);
        if (slot >= 0) buf[slot] = make_key(v, (uint32_t)(c * hw) + ((uint32_t)ent & 0x7fffffffu));
    };
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nseg = min(__builtin_amdgcn_readfirstlane(seg_fill[wave]), STREAM_SEG);   // (written by this wave's lane 0: LDS is in order per wave)
        for (int i = lane; i < nseg; i += 64) convert(stage[wave * STREAM_SEG + i], mxs[wave * STREAM_SEG + i]);
    }
    __syncthreads();
    SD_TRACE(trace0 + 2);
    const Team T{tid, buf, hist, misc, nullptr, stage, 0, alive};
    const int np2k = max(next_pow2(k), 2);
    uint64_t* out = stage1 + (int64_t)blockIdx.x * kmax;
    if (counts[0] <= STREAM_SEG && counts[1] <= STREAM_EXTRA) {         // (block-uniform) nothing overflowed
        for (int i = tid; i < counts[1]; i += STREAM_THREADS) convert(extra[i], extra_mx[i]);
        __syncthreads();
        SD_TRACE(trace0 + 3);
        const int n = counts[2];
        if (n <= k) {                                                   // every candidate is selected: sort them all
            const int np2 = max(next_pow2(n), 2);
            for (int i = n + tid; i < np2; i += STREAM_THREADS) buf[i] = 0ull;
            __syncthreads();
            if (np2 <= RANK_CAP) rank_sort_desc<STREAM_THREADS>(T, buf, np2);
            else if (np2 <= STREAM_THREADS) reg_bitonic_desc<STREAM_THREADS>(T, buf, np2);
            else bitonic_desc<STREAM_THREADS>(T, buf, np2);
            for (int i = tid; i < k; i += STREAM_THREADS) out[i] = i < n ? buf[i] : 0ull;
        } else {
            radix_select_sorted<STREAM_THREADS>(T, LdsSrc<STREAM_THREADS>{buf, n}, k, stage, np2k);
            SD_TRACE(trace0 + 4);
            for (int i = tid; i < k; i += STREAM_THREADS) out[i] = stage[i];
        }
        SD_TRACE(trace0 + 5);
        SD_TRACE(blockIdx.x < 592 ? 7600 + (int)blockIdx.x : -1);
        return;
    }
    // more candidates than the stage holds: second walk, keys straight to the map's global list, selection from there
    __syncthreads();
    if (tid == 0) counts[0] = 0;
    __syncthreads();
    uint64_t* glist = cand + (int64_t)bm * hw + (int64_t)band_lo * ROWS * w;      // this part's rows of the map's h * w slots
    stream_map<true, STREAM_THREADS, ROWS, HALF>(plane, h, w, c, min_score, min_logit, nullptr, nullptr, counts, glist, nullptr, nullptr, nullptr, band_lo, band_hi);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n = counts[0];
    radix_select_sorted<STREAM_THREADS>(T, FlatSrc<STREAM_THREADS, true>{glist, n}, k, buf, np2k);
    for (int i = tid; i < k; i += STREAM_THREADS) out[i] = i < min(n, k) ? buf[i] : 0ull;
}

// A logit below which clamped_sigmoid(x) < min_score FOR SURE (host side of the streaming kernel's early score cut): the exact logit
// of min_score lowered by 1 % (at least 0.01) -- the sigmoid moves by >= 15 ulp over that distance anywhere below 0.999 --, -inf where
// the clamp (1e-6) or a zero threshold lets every pixel pass.  The exact `sigmoid >= min_score` test still runs on what passes.
static float conservative_min_logit(float min_score) {
    if (!(min_score > 1e-5f)) return -INFINITY;
    const double m = std::min((double)min_score, 0.999);
    const double t = std::log(m / (1.0 - m));
    return (float)(t - 1e-2 * std::max(1.0, std::fabs(t)));
}

constexpr int MAP_TILES_MAX = 1024;          // tiles of one map the per-map selector indexes in LDS (2048 x 2048 output maps at 64 x 32 tiles)

// stage 1 (after k_nms_slots; k_map_stream_select does both at once): one block per (image, map): top-k of the map's candidates
// (k = K for anchor maps, P for part maps), sorted, zero padded to k, into stage1[(b * C + m) * kmax ...]
__global__ __launch_bounds__(SEL_THREADS) void k_select_map(const uint64_t* __restrict__ cand, const int* __restrict__ tile_cnt, int tiles,
                                                             int cap, int M, int N, int K, int P, uint64_t* __restrict__ stage1) {
    __shared__ uint64_t buf[SORT_CAP];
    __shared__ uint64_t outb[SD_MAX_TOPK];
    __shared__ int hist[2 * 256];
    __shared__ int misc[4];
    __shared__ int alive[2];
    __shared__ int tcnt[MAP_TILES_MAX], toff[MAP_TILES_MAX + 1];
    __shared__ int wave_tot[SEL_THREADS / 64];
    const int tid = threadIdx.x;
    const int C = M + N;
    const int bm = blockIdx.x, m = bm % C;
    const int k = m < M ? K : P, kmax = max(K, P);
    const int* cnt_g = tile_cnt + (int64_t)bm * tiles;
    [[maybe_unused]] const int trace0 = bm == 0 ? 6100 : (bm == M ? 6200 : -100);
    SD_TRACE(trace0 + 0);
    if (tid < 2) alive[tid] = 0;
    {   // exclusive prefix of the map's tile counts: thread t owns a contiguous chunk (tiles <= MAP_TILES_MAX = 2 * SEL_THREADS)
        const int chunk = (tiles + SEL_THREADS - 1) / SEL_THREADS;
        const int lo = min(tid * chunk, tiles), hi = min(lo + chunk, tiles);
        int sum = 0;
        for (int t = lo; t < hi; ++t) { const int v = cnt_g[t]; tcnt[t] = v; sum += v; }
        int incl = sum;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wave; ++q) before += wave_tot[q];
        int run = before + incl - sum;
        for (int t = lo; t < hi; ++t) { toff[t] = run; run += tcnt[t]; }
        if (tid == SEL_THREADS - 1) toff[tiles] = run;
    }
    __syncthreads();
    const int n = toff[tiles];
    const Team T{tid, buf, hist, misc, nullptr, outb, 0, alive};
    // tile0 = SPEC_TILES: no tile of this source has speculative keys (that is the one-launch kernel's hand-off record)
    const TiledSrc<SEL_THREADS, false> src{cand + (int64_t)bm * tiles * cap, tcnt, toff, nullptr, SPEC_TILES, tiles, n, cap};
    SD_TRACE(trace0 + 1);
    team_select_topk<SEL_THREADS>(T, src, k, n, k, SORT_CAP);
    SD_TRACE(trace0 + 2);
    const int take = min(n, k);
    for (int i = tid; i < k; i += SEL_THREADS) stage1[(int64_t)bm * kmax + i] = i < take ? buf[i] : 0ull;      // zero padded: no counts downstream
    SD_TRACE(trace0 + 3);
}

// stage 2: one block per (image, map) again.  The stage-1 lists are sorted (and zero padded to k: a zero key is smaller than every
// candidate's), so the final rank of a key is its position in its own list plus, for every other list of its group, the number of keys
// there that beat it (binary search; keys are unique): no radix select, no sort, and B x (M + N) blocks instead of one block per
// image.  Keys with rank < k land at final[(b * 2 + group) * kmax + rank] -- the group's top-k in the reference's order (utils.py:459);
// the block of the group's first map zeroes the slots beyond the group's candidates (the ranks are dense: 0 .. candidates - 1).
constexpr int RANK_KEYS_MAX = 8192;         // keys of one group (maps x k) held in LDS by k_rank_maps: 64 KB
__global__ __launch_bounds__(SEL_THREADS) void k_rank_maps(const uint64_t* __restrict__ stage1, int M, int N, int K, int P,
                                                            uint64_t* __restrict__ final_keys) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int total;
    const int tid = threadIdx.x;
    const int C = M + N, kmax = max(K, P);
    const int bm = blockIdx.x, b = bm / C, m = bm - b * C;
    const int grp = m >= M ? 1 : 0;
    const int nl = grp ? N : M, first = grp ? M : 0, own = m - first, k = grp ? P : K;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);        // [nl][k]
    [[maybe_unused]] const int trace0 = bm == 0 ? 6500 : (bm == M ? 6510 : (bm == (int)gridDim.x - 1 ? 6520 : -100));
    SD_TRACE(trace0 + 0);
    if (tid == 0) total = 0;
    __syncthreads();
    const uint64_t* src = stage1 + ((int64_t)b * C + first) * kmax;
    const int span = nl * k;
    int nonzero = 0;
    for (int base = tid; base < span; base += 8 * SEL_THREADS) {            // every load of the thread in flight together
        uint64_t kv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * SEL_THREADS;
            const int l = i / k;
            kv[u] = i < span ? src[(int64_t)l * kmax + (i - l * k)] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (base + u * SEL_THREADS < span) { keys[base + u * SEL_THREADS] = kv[u]; nonzero += kv[u] != 0ull; }
    }
    if (own == 0) wave_atomic_add(&total, nonzero);                          // (block-uniform condition)
    __syncthreads();
    SD_TRACE(trace0 + 1);
    int top = 1;                                                // largest power of two <= k: first probe of the branch-free search
    while (top * 2 <= k) top *= 2;
    uint64_t* out = final_keys + ((int64_t)b * 2 + grp) * kmax;
    for (int i = tid; i < k; i += SEL_THREADS) {
        const uint64_t key = keys[own * k + i];
        if (key == 0ull) continue;
        int rank = i;
        for (int l0 = 0; l0 < nl; l0 += 4) {                    // four lists at a time: their (dependent) probes overlap
            int pos[4] = {0, 0, 0, 0};
            const uint64_t* lst[4];
            bool on[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                on[u] = l0 + u < nl && l0 + u != own;
                lst[u] = keys + (on[u] ? l0 + u : own) * k;     // descending; count of keys greater than `key`
            }
            for (int step = top; step > 0; step >>= 1) {        // branch-free: four probes in flight, then four updates
                uint64_t probe[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) probe[u] = lst[u][min(pos[u] + step, k) - 1];
#pragma unroll
                for (int u = 0; u < 4; ++u) pos[u] += (pos[u] + step <= k && probe[u] > key) ? step : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) rank += on[u] ? pos[u] : 0;
        }
        if (rank < k) out[rank] = key;
    }
    SD_TRACE(trace0 + 2);
    if (own == 0)
        for (int i = min(total, k) + tid; i < k; i += SEL_THREADS) out[i] = 0ull;
}

// stage 3: association, GROUP_PARTS parts per block (grid: B x ceil(P / GROUP_PARTS)), four lanes per part.  Every block decodes the
// image's K anchors (zero slots filled as fill_zero_slots does) and refines them (K gathers); block 0 of an image also writes the
// anchor outputs.  A part's four lanes scan interleaved quarters of the anchors with score > conf (a prefix: the list is sorted) and
// combine by (distance, anchor rank): the reference's first minimum over the rounded distances (decoders.py:88-100).
constexpr int GROUP_PARTS = 64, GROUP_THREADS = 256;
// keys[npos..k) := the zero slots of the reference's top-k on a map whose suppressed pixels are exactly 0 (utils.py:451): the lowest
// class-major flat indices that are not peaks, ascending (same rule as fill_zero_slots).  Block-wide, GROUP_THREADS threads; `flags` has
// k ints.  Only works when the list has fewer than k peaks; always executes the same four barriers.
template <int NT = GROUP_THREADS>
__device__ void fill_zero_keys(uint64_t* keys, int npos, int k, int* flags, int* wave_tot) {
    constexpr int GROUP_THREADS = NT;                             // (shadows the namespace constant: the body is written against it)
    const int tid = threadIdx.x;
    const bool need = npos < k;
    if (need)
        for (int f = tid; f < k; f += GROUP_THREADS) flags[f] = 1;
    __syncthreads();
    if (need)
        for (int j = tid; j < npos; j += GROUP_THREADS) {          // a peak whose flat index is below k takes that index out
            const uint32_t flat = (uint32_t)(~keys[j]);
            if (flat < (uint32_t)k) flags[flat] = 0;
        }
    __syncthreads();
    const int per = (k + GROUP_THREADS - 1) / GROUP_THREADS;     // thread t owns flat indices [t * per, t * per + per)
    const int lo = min(tid * per, k), hi = min(lo + per, k);
    int sum = 0;
    if (need)
        for (int f = lo; f < hi; ++f) sum += flags[f];
    int incl = sum;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (need) {
        int run = incl - sum;
        for (int q = 0; q < wave; ++q) run += wave_tot[q];
        for (int f = lo; f < hi; ++f)
            if (flags[f]) {
                if (npos + run < k) keys[npos + run] = make_key(0.0f, (uint32_t)f);
                ++run;
            }
    }
    __syncthreads();
}

// fill_zero_keys for the anchor list and the part list of an image in the SAME four barriers (k_group_wide: the two calls in sequence were
// 2 us of a 9.5 us kernel at the stress shape -- eight barriers with a dependent LDS round trip between each pair).  `flags` has ka + kb ints.
__device__ void fill_zero_keys_pair(uint64_t* keys_a, int npos_a, int ka, uint64_t* keys_b, int npos_b, int kb, int* flags, int (*wave_tot)[GROUP_THREADS / 64]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t* const keys[2] = {keys_a, keys_b};
    const int npos[2] = {npos_a, npos_b}, k[2] = {ka, kb};
    int* const fl[2] = {flags, flags + ka};
    const bool need[2] = {npos_a < ka, npos_b < kb};
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (need[g])
            for (int f = tid; f < k[g]; f += GROUP_THREADS) fl[g][f] = 1;
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (need[g])
            for (int j = tid; j < npos[g]; j += GROUP_THREADS) {      // a peak whose flat index is below k takes that index out
                const uint32_t flat = (uint32_t)(~keys[g][j]);
                if (flat < (uint32_t)k[g]) fl[g][flat] = 0;
            }
    __syncthreads();
    int lo[2], hi[2], sum[2], incl[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int per = (k[g] + GROUP_THREADS - 1) / GROUP_THREADS;  // thread t owns flat indices [t * per, t * per + per)
        lo[g] = min(tid * per, k[g]); hi[g] = min(lo[g] + per, k[g]);
        sum[g] = 0;
        if (need[g])
            for (int f = lo[g]; f < hi[g]; ++f) sum[g] += fl[g][f];
        incl[g] = sum[g];
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v0 = __shfl_up(incl[0], o), v1 = __shfl_up(incl[1], o);
        if (lane >= o) { incl[0] += v0; incl[1] += v1; }
    }
    if (lane == 63) { wave_tot[0][wave] = incl[0]; wave_tot[1][wave] = incl[1]; }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g)
        if (need[g]) {
            int run = incl[g] - sum[g];
            for (int q = 0; q < wave; ++q) run += wave_tot[g][q];
            for (int f = lo[g]; f < hi[g]; ++f)
                if (fl[g][f]) {
                    if (npos[g] + run < k[g]) keys[g][npos[g] + run] = make_key(0.0f, (uint32_t)f);
                    ++run;
                }
        }
    __syncthreads();
}

__global__ __launch_bounds__(GROUP_THREADS) void k_group_wide(const uint64_t* __restrict__ final_keys, int h, int w, int K, int P, float conf,
                                                               float dist_px, RegMaps rm, void* packed, int B) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* akeys = reinterpret_cast<uint64_t*>(smem);                 // [K]
    uint64_t* pkeys = akeys + K;                                          // [P]
    const int Kp = (K + 3) & ~3;                                          // (float4 reads of the anchor positions)
    float* posx = reinterpret_cast<float*>(pkeys + P);                    // [Kp]
    float* posy = posx + Kp;                                              // [Kp]
    int* flags = reinterpret_cast<int*>(posy + Kp);                       // [K + P]
    __shared__ int n_live_s, cnt_s[2];
    __shared__ int wave_tot[2][GROUP_THREADS / 64];
    const int tid = threadIdx.x;
    const int b = blockIdx.x, chunk = blockIdx.y;
    const int hw = h * w, kmax = max(K, P);
    const uint64_t* fa = final_keys + ((int64_t)b * 2 + 0) * kmax;
    const uint64_t* fp = final_keys + ((int64_t)b * 2 + 1) * kmax;
    [[maybe_unused]] const int trace0 = (b == 0 && chunk == 0) ? 6600 : ((b == (int)gridDim.x - 1 && chunk == (int)gridDim.y - 1) ? 6610 : -100);
    SD_TRACE(trace0 + 0);
    if (tid == 0) { n_live_s = 0; cnt_s[0] = 0; cnt_s[1] = 0; }
    // every load that does not depend on another goes out first: this thread's anchor keys, its part key, and the LAST key of both
    // lists (zero <=> the list has fewer candidates than slots; the zeros are a suffix); the chain is keys -> gathers -> scan
    const int p = chunk * GROUP_PARTS + (tid >> 2), q = tid & 3;
    uint64_t pkey = (p < P) ? fp[p] : 0ull;
    constexpr int APT = SD_MAX_TOPK / GROUP_THREADS;            // anchor keys per thread
    uint64_t ak[APT];
#pragma unroll
    for (int u = 0; u < APT; ++u) ak[u] = (tid + u * GROUP_THREADS < K) ? fa[tid + u * GROUP_THREADS] : 0ull;
    // (the whole part list too, speculatively: a short list needs it in LDS for its zero slots, and asking for it only once `p_last` has
    //  arrived is one more dependent round trip -- the annotations-only mode, where lists are mostly padding, always took it)
    uint64_t pk[APT];
#pragma unroll
    for (int u = 0; u < APT; ++u) pk[u] = (tid + u * GROUP_THREADS < P) ? fp[tid + u * GROUP_THREADS] : 0ull;
    const uint64_t a_last = fa[K - 1], p_last = fp[P - 1];
    const PackedLayout L = packed_layout(packed, B, K, P);
    const float* off_b = rm.offsets + (int64_t)b * rm.o_sb;
    const float* emb_b = rm.embeddings + (int64_t)b * rm.e_sb;
    if (a_last == 0ull || p_last == 0ull) {                     // (block-uniform) fewer peaks than slots: the lists go through LDS for the zero slots
        __syncthreads();
        // (the peaks are counted per wave by ballots, one LDS atomic per wave and list: 256 lanes adding to ONE counter serialise --
        //  traced 4.7 us from the block's start to this barrier at the stress shape, 1.7 for the loads alone)
        int na_w = 0, np_w = 0;
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const bool ina = tid + u * GROUP_THREADS < K, inp = tid + u * GROUP_THREADS < P;
            if (ina) akeys[tid + u * GROUP_THREADS] = ak[u];
            if (inp) pkeys[tid + u * GROUP_THREADS] = pk[u];
            na_w += __popcll(__ballot(ina && ak[u] != 0ull));
            np_w += __popcll(__ballot(inp && pk[u] != 0ull));
        }
        if ((tid & 63) == 0) {
            if (na_w) atomicAdd(&cnt_s[0], na_w);
            if (np_w) atomicAdd(&cnt_s[1], np_w);
        }
        __syncthreads();
        const int na = cnt_s[0], np = cnt_s[1];
        SD_TRACE(trace0 + 4);
        fill_zero_keys_pair(akeys, na, K, pkeys, np, P, flags, wave_tot);
        SD_TRACE(trace0 + 5);
#pragma unroll
        for (int u = 0; u < APT; ++u)
            if (tid + u * GROUP_THREADS < K) ak[u] = akeys[tid + u * GROUP_THREADS];
        if (p < P) pkey = pkeys[p];
    }
    // this thread's part: its gathers are issued before the anchors' pass (they do not depend on it)
    float ex = 0.f, ey = 0.f, pox = 0.f, poy = 0.f;
    int p_ind = 0, p_cls = 0;
    if (p < P) {
        const uint32_t flat = ~(uint32_t)pkey;
        p_cls = flat / hw; p_ind = flat - p_cls * hw;
        ex = emb_b[p_ind]; ey = emb_b[rm.e_sc + p_ind];                                // decoders.py:66
        pox = off_b[p_ind]; poy = off_b[rm.o_sc + p_ind];
    }
    int last_live = 0;
#pragma unroll
    for (int u = 0; u < APT; ++u) {
        const int a = tid + u * GROUP_THREADS;
        if (a >= K) break;
        const uint64_t key = ak[u];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw, ind = flat - cls * hw;
        const int y = ind / w, x = ind - y * w;
        const float score = ord2f((uint32_t)(key >> 32));
        const float ax = (float)x + off_b[ind];                 // decoders.py:52
        const float ay = (float)y + off_b[rm.o_sc + ind];       // decoders.py:53
        const bool mk = score > conf;                           // decoders.py:83
        if (mk) last_live = a + 1;
        posx[a] = mk ? ax : 1e6f;                               // decoders.py:85-86
        posy[a] = mk ? ay : 1e6f;
        if (chunk == 0) {
            float* ao = L.anchor_out + ((int64_t)b * K + a) * 4;
            ao[0] = ax; ao[1] = ay; ao[2] = score; ao[3] = (float)cls;
            L.anchor_smask[(int64_t)b * K + a] = mk ? score : -1.0f; // decoders.py:84
            L.anchor_ind[(int64_t)b * K + a] = ind;
        }
    }
    wave_atomic_max(&n_live_s, last_live);
    SD_TRACE(trace0 + 1);
    lds_barrier();                                              // (not __syncthreads(): block 0's anchor stores stay in flight)
    SD_TRACE(trace0 + 2);
    // anchors beyond the last live rank are all masked: at (1e6, 1e6) they are never within dist_px of a live part (see block_group)
    const int n_scan = dist_px < 1e5f ? n_live_s : K;
    if (p < P) {
        const int cls = p_cls, ind = p_ind;
        const int y = ind / w, x = ind - y * w;
        const float score = ord2f((uint32_t)(pkey >> 32));
        const float px = (float)x + pox;                                                // decoders.py:67
        const float py = (float)y + poy;                                                // decoders.py:68
        const float ox = px + ex, oy = py + ey;                 // decoders.py:69-70
        const bool mk = score > conf;                           // decoders.py:78
        const float orx = mk ? ox : -1e6f, ory = mk ? oy : -1e6f; // decoders.py:80-81
        float best = INFINITY;
        int best_a = 0x7fffffff;
        if (mk || dist_px >= 1e5f) {
            // decoders.py:88-98, utils.py:433-435.  Lane q scans the anchors of its quarter (whole groups of four: one ds_read_b128
            // of x and of y per four anchors), ascending: first minimum within the lane
            const int groups = (n_scan + 3) >> 2, per = (groups + 3) >> 2;
            const int g_lo = q * per, g_hi = min(g_lo + per, groups);
            for (int gi = g_lo; gi < g_hi; ++gi) {
                const float4 ax4 = reinterpret_cast<const float4*>(posx)[gi], ay4 = reinterpret_cast<const float4*>(posy)[gi];
                const float axs[4] = {ax4.x, ax4.y, ax4.z, ax4.w}, ays[4] = {ay4.x, ay4.y, ay4.z, ay4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dx = orx - axs[e], dy = ory - ays[e];
                    const float sx = dx * dx, sy = dy * dy;
                    const float d = sqrtf(sx + sy);
                    if (d < best && gi * 4 + e < n_scan) { best = d; best_a = gi * 4 + e; }
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 4; o <<= 1) {                       // the part's four lanes: minimum by (distance, anchor rank)
            const float od = __shfl_xor(best, o);
            const int oa = __shfl_xor(best_a, o);
            if (od < best || (od == best && oa < best_a)) { best = od; best_a = oa; }
        }
        if (q == 0) {
            float* po = L.part_out + ((int64_t)b * P + p) * 6;
            po[0] = px; po[1] = py; po[2] = score; po[3] = (float)cls; po[4] = ox; po[5] = oy;
            L.part_emb[((int64_t)b * P + p) * 2 + 0] = ex;
            L.part_emb[((int64_t)b * P + p) * 2 + 1] = ey;
            L.part_smask[(int64_t)b * P + p] = mk ? score : -1.0f;   // decoders.py:79
            L.part_ind[(int64_t)b * P + p] = ind;
            L.assign[(int64_t)b * P + p] = (best < dist_px) ? best_a : -1;   // decoders.py:100
        }
    }
    SD_TRACE(trace0 + 3);
    if (chunk == 0 && tid == 0) L.status[b] = 0;
}

// stages 2 + 3 in ONE launch (round 5), one block per image: the stage-1 lists of both groups (maps x parts-of-a-map lists of k sorted,
// zero padded keys) are loaded into LDS once, every key finds its final rank like in k_rank_maps (its position in its own list + the
// number of keys that beat it in every other list of its group: branch-free binary searches) and drops into the group's final list in
// LDS; zero slots are filled like fill_zero_slots does and block_group runs the gathers + the K x P association (decoders.py:49-100) --
// the same device functions, the same values as k_rank_maps + k_group_wide, without the launch between them and without the trip of the
// final lists through global memory (two dependent round trips less on a chain that is nothing but dependent round trips).  The
// round-3 one-block-per-image selector was slow because it SELECTED per image; here the per-map selection is already done and an image
// is a few thousand keys at most.
template <int NT>
__global__ __launch_bounds__(NT) void k_rank_group(const uint64_t* __restrict__ stage1, int LM, int LN, int h, int w, int K, int P, float conf,
                                                    float dist_px, RegMaps rm, void* packed, int B) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                  // [LM * K] anchors' lists, then [LN * P] parts' lists
    const int na_keys = LM * K, np_keys = LN * P, n_keys = na_keys + np_keys;
    uint64_t* fin = keys + n_keys;                                        // [K] final anchor keys, [P] final part keys
    __shared__ float as_[SD_MAX_TOPK], ps_[SD_MAX_TOPK], posx[SD_MAX_TOPK], posy[SD_MAX_TOPK];
    __shared__ int ai_[SD_MAX_TOPK], ac_[SD_MAX_TOPK], pi_[SD_MAX_TOPK], pc_[SD_MAX_TOPK];
    __shared__ int flags[SD_MAX_TOPK];
    __shared__ int wave_tot[NT / 64];
    __shared__ int cnt_s[2];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int hw = h * w, kmax = max(K, P);
    const uint64_t* src = stage1 + (int64_t)b * (LM + LN) * kmax;
    [[maybe_unused]] const int trace0 = b == 0 ? 6700 : (b == (int)gridDim.x - 1 ? 6710 : -100);
    SD_TRACE(trace0 + 0);
    if (tid < 2) cnt_s[tid] = 0;
    for (int base = tid; base < n_keys; base += 4 * NT) {               // every load of the thread in flight together
        uint64_t kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT;
            const bool part = i >= na_keys;
            const int j = part ? i - na_keys : i, k = part ? P : K;
            const int l = j / k;
            kv[u] = i < n_keys ? src[(int64_t)((part ? LM : 0) + l) * kmax + (j - l * k)] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * NT < n_keys) keys[base + u * NT] = kv[u];
    }
    for (int i = tid; i < K + P; i += NT) fin[i] = 0ull;
    __syncthreads();
    SD_TRACE(trace0 + 1);
    int nz[2] = {0, 0};
    for (int idx = tid; idx < n_keys; idx += NT) {
        const uint64_t key = keys[idx];
        if (key == 0ull) continue;
        const int grp = idx >= na_keys ? 1 : 0;
        const int k = grp ? P : K, nl = grp ? LN : LM;
        const uint64_t* lists = keys + (grp ? na_keys : 0);
        const int j = idx - (grp ? na_keys : 0);
        const int own = j / k;
        int rank = j - own * k;
        int top = 1;                                            // largest power of two <= k: first probe of the branch-free search
        while (top * 2 <= k) top *= 2;
        for (int l = 0; l < nl; ++l) {
            if (l == own) continue;
            const uint64_t* lst = lists + l * k;                // descending, zero padded: count of keys greater than `key`
            int pos = 0;
            for (int step = top; step > 0; step >>= 1) {
                const uint64_t probe = lst[min(pos + step, k) - 1];
                pos += (pos + step <= k && probe > key) ? step : 0;
            }
            rank += pos;
        }
        ++nz[grp];
        if (rank < k) fin[(grp ? K : 0) + rank] = key;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { nz[0] += __shfl_xor(nz[0], o); nz[1] += __shfl_xor(nz[1], o); }   // one LDS atomic per wave and list
    if ((tid & 63) == 0) {
        if (nz[0]) atomicAdd(&cnt_s[0], nz[0]);
        if (nz[1]) atomicAdd(&cnt_s[1], nz[1]);
    }
    __syncthreads();
    SD_TRACE(trace0 + 2);
    const int na = min(cnt_s[0], K), np = min(cnt_s[1], P);     // the ranks are dense: the first min(candidates, k) slots are taken
    fill_zero_keys<NT>(fin, na, K, flags, wave_tot);
    fill_zero_keys<NT>(fin + K, np, P, flags, wave_tot);
    for (int i = tid; i < K + P; i += NT) {
        const uint64_t key = fin[i];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw;
        const bool part = i >= K;
        const int o = part ? i - K : i;
        (part ? ps_ : as_)[o] = ord2f((uint32_t)(key >> 32));
        (part ? pi_ : ai_)[o] = flat - cls * hw;
        (part ? pc_ : ac_)[o] = cls;
    }
    __syncthreads();
    SD_TRACE(trace0 + 3);
    const PackedLayout L = packed_layout(packed, B, K, P);
    block_group(b, K, P, w, conf, dist_px, rm, as_, ai_, ac_, ps_, pi_, pc_, posx, posy, L);
    SD_TRACE(trace0 + 4);
}

// k_rank_group for SMALL selections (K, P <= 64, <= 512 stage-1 keys per group: the cfg shape, 2 + 1 maps, K = 20, P = 40, maps split in
// three): the chain of an image is nothing but dependent round trips, so every one that can be taken off it is:
//   * a thread that owns a stage-1 key issues that key's gathers (offset x / y, and embedding x / y for a part) the moment the key has
//     arrived -- BEFORE the ranks exist: the gathers travel with the key to its final slot (4 floats through LDS) instead of being a second
//     round trip behind the ranking;
//   * the zero slots of a short list (utils.py:451: the lowest class-major flat indices that are not peaks; always < k <= 64, class 0) read
//     their gathers from the first 64 pixels of the four regression planes, fetched at kernel start;
//   * the rank of a key = its position in its own list + the number of GREATER keys in the other lists of its group, counted over the
//     group's keys with independent 16-byte LDS reads (two keys each; every lane reads the same address: a broadcast) -- a binary search
//     is ~6 dependent LDS round trips per list (traced: 4.1 us for nine lists);
//   * zero slots by one wave per group with a ballot (fill_zero_slots' rule), both groups side by side.
// Same values as k_rank_maps + k_group_wide / block_group: the arithmetic of decoders.py:49-100 is restated operation by operation.
constexpr int RGS_THREADS = 256, RGS_KPT = 4, RGS_KEYS = 512, RGS_K = 64;
__global__ __launch_bounds__(RGS_THREADS) void k_rank_group_small(const uint64_t* __restrict__ stage1, int LM, int LN, int h, int w, int K, int P,
                                                                   float conf, float dist_px, RegMaps rm, void* packed, int B) {
    __shared__ __align__(16) uint64_t keys[2][RGS_KEYS + 2];             // the group's lists, padded with a zero key to an even count
    __shared__ float zpre[4][RGS_K];                                      // offsets x / y, embeddings x / y of the pixels 0 .. 63
    __shared__ uint64_t fkey[2][RGS_K];                                   // final keys by rank
    __shared__ float fg[2][4][RGS_K];                                     // their gathers by rank
    __shared__ float posx[RGS_K], posy[RGS_K];
    __shared__ int cnt_s[2], n_live_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int hw = h * w, kmax = max(K, P);
    const int na_keys = LM * K, np_keys = LN * P, n_keys = na_keys + np_keys;
    const uint64_t* src = stage1 + (int64_t)b * (LM + LN) * kmax;
    const float* off_b = rm.offsets + (int64_t)b * rm.o_sb;
    const float* emb_b = rm.embeddings + (int64_t)b * rm.e_sb;
    [[maybe_unused]] const int trace0 = b == 0 ? 6700 : (b == (int)gridDim.x - 1 ? 6710 : -100);
    SD_TRACE(trace0 + 0);
    // ---- round trip 1: this thread's keys, and the regression values of the pixels a zero slot can name
    uint64_t kv[RGS_KPT];
#pragma unroll
    for (int u = 0; u < RGS_KPT; ++u) {
        const int i = tid + u * RGS_THREADS;
        const bool part = i >= na_keys;
        const int j = part ? i - na_keys : i, k = part ? P : K;
        const int l = j / k;
        kv[u] = i < n_keys ? src[(int64_t)((part ? LM : 0) + l) * kmax + (j - l * k)] : 0ull;
    }
    {
        const int plane = tid >> 6, f = min(lane, hw - 1);               // wave q fetches plane q
        const float* pl = plane == 0 ? off_b : (plane == 1 ? off_b + rm.o_sc : (plane == 2 ? emb_b : emb_b + rm.e_sc));
        zpre[plane][lane] = pl[f];
    }
    if (tid < 2) cnt_s[tid] = 0;
    if (tid == 2) n_live_s = 0;
    for (int i = tid; i < 2 * RGS_K; i += RGS_THREADS) fkey[i / RGS_K][i % RGS_K] = 0ull;
    __syncthreads();                                                     // the counters are zero before any wave allocates from them
    // ---- round trip 2 (in flight while the ranks are counted): the gathers of every key this thread owns
    // The group's NON-ZERO keys are compacted into keys[grp] (any order: they are only counted over; slots by one LDS atomic per wave
    // and group): the zero padding of a short list is never ahead of a peak, and in the annotations-only mode a list is mostly padding
    // (cfg shape: ~10 of 80 keys per group) -- the counting loop below is a chain of dependent LDS reads, ~1.5 us over 80 keys.
    float gv[RGS_KPT][4];
#pragma unroll
    for (int u = 0; u < RGS_KPT; ++u) {
        const int i = tid + u * RGS_THREADS;
        const bool part = i >= na_keys;
        const bool live = i < n_keys && kv[u] != 0ull;
        gv[u][0] = gv[u][1] = gv[u][2] = gv[u][3] = 0.f;
        if (live) {
            const uint32_t flat = ~(uint32_t)kv[u];
            const int ind = (int)(flat % (uint32_t)hw);
            gv[u][0] = off_b[ind]; gv[u][1] = off_b[rm.o_sc + ind];
            if (part) { gv[u][2] = emb_b[ind]; gv[u][3] = emb_b[rm.e_sc + ind]; }
        }
        const int sa = alloc_slot(&cnt_s[0], live && !part), sp = alloc_slot(&cnt_s[1], live && part);
        if (live) keys[part ? 1 : 0][part ? sp : sa] = kv[u];
    }
    __syncthreads();
    SD_TRACE(trace0 + 1);
    // ---- ranks by counting
#pragma unroll
    for (int u = 0; u < RGS_KPT; ++u) {
        const int i = tid + u * RGS_THREADS;
        const uint64_t key = kv[u];
        if (i >= n_keys || key == 0ull) continue;
        const int grp = i >= na_keys ? 1 : 0;
        const int n = cnt_s[grp], k = grp ? P : K;
        const ulonglong2* pairs = reinterpret_cast<const ulonglong2*>(keys[grp]);
        int rank = 0;                                                    // keys are unique: `>` counts exactly the keys ahead of this one
        for (int q = 0; q < n / 2; ++q) {                                // (kept rolled: a one-shot kernel runs from a cold instruction cache --
            const ulonglong2 two = pairs[q];                             //  unrolled by eight it took 4.4 instead of 2.5 us on 120 keys)
            rank += (two.x > key ? 1 : 0) + (two.y > key ? 1 : 0);
        }
        if (n & 1) rank += keys[grp][n - 1] > key ? 1 : 0;
        if (rank < k) {
            fkey[grp][rank] = key;
            fg[grp][0][rank] = gv[u][0]; fg[grp][1][rank] = gv[u][1]; fg[grp][2][rank] = gv[u][2]; fg[grp][3][rank] = gv[u][3];
        }
    }
    __syncthreads();
    SD_TRACE(trace0 + 2);
    // ---- zero slots (fill_zero_slots' one-wave rule), wave 0: anchors, wave 1: parts
    if (wave < 2) {
        const int k = wave ? P : K, npos = min(cnt_s[wave], k);
        if (npos < k) {
            int used = 0;
            for (int j = 0; j < npos; ++j) used |= ((uint32_t)(~fkey[wave][j]) == (uint32_t)lane);
            const bool flag = lane < k && !used;
            const unsigned long long m = __ballot(flag);
            const int slot = npos + __popcll(m & ((1ull << lane) - 1ull));
            if (flag && slot < k) {
                fkey[wave][slot] = make_key(0.0f, (uint32_t)lane);
                fg[wave][0][slot] = zpre[0][lane]; fg[wave][1][slot] = zpre[1][lane];
                fg[wave][2][slot] = zpre[2][lane]; fg[wave][3][slot] = zpre[3][lane];
            }
        }
    }
    __syncthreads();
    SD_TRACE(trace0 + 3);
    // ---- decoders.py:49-100 (block_group's arithmetic): anchors on wave 0, then every part on its own thread
    const PackedLayout L = packed_layout(packed, B, K, P);
    if (tid < K) {
        const int a = tid;
        const uint64_t key = fkey[0][a];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw, ind = flat - cls * hw;
        const int y = ind / w, x = ind - y * w;
        const float score = ord2f((uint32_t)(key >> 32));
        const float ax = (float)x + fg[0][0][a];                // decoders.py:52
        const float ay = (float)y + fg[0][1][a];                // decoders.py:53
        const bool mk = score > conf;                           // decoders.py:83
        posx[a] = mk ? ax : 1e6f;                               // decoders.py:85-86
        posy[a] = mk ? ay : 1e6f;
        const unsigned long long lv = __ballot(mk);              // (K <= 64: the anchors are lanes of wave 0; the ballot sees the active ones)
        if (a == 0) n_live_s = lv ? 64 - __clzll((long long)lv) : 0;   // last live rank + 1, no atomic
        float* ao = L.anchor_out + ((int64_t)b * K + a) * 4;
        ao[0] = ax; ao[1] = ay; ao[2] = score; ao[3] = (float)cls;
        L.anchor_smask[(int64_t)b * K + a] = mk ? score : -1.0f; // decoders.py:84
        L.anchor_ind[(int64_t)b * K + a] = ind;
    }
    lds_barrier();                                              // (not __syncthreads(): the anchors' stores stay in flight)
    const int n_live = n_live_s;
    if (tid >= 64 && tid - 64 < P) {                            // (waves 1 ..: the anchors' stores of wave 0 are not in their way)
        const int p = tid - 64;
        const uint64_t key = fkey[1][p];
        const uint32_t flat = ~(uint32_t)key;
        const int cls = flat / hw, ind = flat - cls * hw;
        const int y = ind / w, x = ind - y * w;
        const float score = ord2f((uint32_t)(key >> 32));
        const float ex = fg[1][2][p], ey = fg[1][3][p];         // decoders.py:66
        const float px = (float)x + fg[1][0][p];                // decoders.py:67
        const float py = (float)y + fg[1][1][p];                // decoders.py:68
        const float ox = px + ex, oy = py + ey;                 // decoders.py:69-70
        const bool mk = score > conf;                           // decoders.py:78
        const float orx = mk ? ox : -1e6f, ory = mk ? oy : -1e6f; // decoders.py:80-81
        float best = INFINITY, best_s = INFINITY;               // (see block_group: masked anchors are a suffix and never within dist_px)
        int best_a = 0;
        if (mk && dist_px < 1e5f) {
            for (int a = 0; a < n_live; ++a) {
                const float dx = orx - posx[a], dy = ory - posy[a];
                const float sx = dx * dx, sy = dy * dy;
                const float ss = sx + sy;
                if (ss < best_s) {
                    const float d = sqrtf(ss);
                    best_s = ss;
                    if (d < best) { best = d; best_a = a; }     // strict <: lowest anchor rank wins ties
                }
            }
        } else {
            for (int a = 0; a < K; ++a) {                       // (absurd thresholds: the reference's full scan)
                const float dx = orx - posx[a], dy = ory - posy[a];
                const float sx = dx * dx, sy = dy * dy;
                const float d = sqrtf(sx + sy);
                if (d < best) { best = d; best_a = a; }
            }
        }
        float* po = L.part_out + ((int64_t)b * P + p) * 6;
        po[0] = px; po[1] = py; po[2] = score; po[3] = (float)cls; po[4] = ox; po[5] = oy;
        L.part_emb[((int64_t)b * P + p) * 2 + 0] = ex;
        L.part_emb[((int64_t)b * P + p) * 2 + 1] = ey;
        L.part_smask[(int64_t)b * P + p] = mk ? score : -1.0f;   // decoders.py:79
        L.part_ind[(int64_t)b * P + p] = ind;
        L.assign[(int64_t)b * P + p] = (best < dist_px) ? best_a : -1;   // decoders.py:100
    }
    if (tid == 0) L.status[b] = 0;
#ifdef SD_DECODE_TRACE
    __syncthreads();                                            // (trace builds: the parts' threads are done too)
#endif
    SD_TRACE(trace0 + 4);
}

// ---------------------------------------------------------------------------------------------
// ONE-launch decoder (sd_decode_fused): NMS tile blocks + one SELECTOR block per image in the same grid.  bs = 1 inference is
// latency-bound (199 KB of algorithmic traffic): what counts is the number of DEPENDENT global round trips (~2 us each across
// XCDs) and of same-address atomics (~0.15-0.2 us EACH, serialised at the memory side: a per-image arrival counter bumped by
// every tile block cost 7 us at 48 tiles and 200 us at 1024 -- measured, first version of this kernel).  So:
//   * no second launch, no counter memset, no atomic read-modify-write anywhere;
//   * a tile block's chain is load -> NMS -> stores: every tile owns TILE_CAP candidate slots in the scratch workspace and a
//     64-byte hand-off RECORD {valid | count, first SPEC keys} in the caller's zero-initialised state buffer;
//   * the selector blocks are the LAST blocks of the (linear) grid, so every tile block has been dispatched before a selector
//     starts to wait (at most B selectors spin; the launcher caps B far below the number of resident block slots);
//     thread t polls record t (one round trip: count word + 6 keys), takes a record when the valid bit is set and the keys it
//     announces are non-zero (self-validating granules: guide, Guideline 16 R2 -- the loads of one poll are not ordered among
//     themselves), zeroes it again (state left zero: back-to-back calls and hipGraph replays need no memset), fetches the keys
//     beyond SPEC from the tile's slots (published BEFORE the record: sc1 stores, s_waitcnt vmcnt(0), barrier, then the
//     record's count word), sorts in LDS, gathers offsets / embeddings, writes the packed result.
// All hand-off stores / loads are agent-scope relaxed atomics (sc1: written through, never read from a stale L1 line).
// A bounded spin: after ~2^21 polls a selector gives up and reports status 1 for its image instead of hanging the GPU.
// 256 threads; dynamic LDS sized by the host from K, P and the tile count (cfg: ~22 KB).
// Results are bit-identical to sd_decode (same keys, same total order, same block_group arithmetic).
// ---------------------------------------------------------------------------------------------
constexpr int FUSED_THREADS = 256;
constexpr int FUSED_TEAM = 128;               // the selector works on the anchor list and the part list side by side: 2 teams x 2 waves
// keys per team sorted in LDS (longer lists take the radix select), chosen by the launcher: the LDS of the selector is paid by
// every tile block of the grid (occupancy), and the two selection modes see very different list lengths -- annotations-only:
// only peaks >= conf (tens); exact top-k: every NMS survivor (~4 % of all pixels on noisy maps)
constexpr int FUSED_CAP_FAST = 512, FUSED_CAP_EXACT = 2048;
constexpr int FUSED_MAX_TOPK = 512;
constexpr int REC_WORDS = 16;                 // one 64-byte record per tile: word 0 = 0x80000000 | count, bytes 8..55 = SPEC keys
constexpr unsigned REC_VALID = 0x80000000u;
constexpr int POLL_LIMIT = 1 << 21;

struct FusedLds {       // byte offsets into the dynamic LDS block of the selector
    int buf, spec, tcnt, toff, hist, flags, out, outk, as_, ps_, posx, posy, ai_, ac_, pi_, pc_, total;
};
__host__ __device__ inline FusedLds fused_lds(int K, int P, int ntiles_img, int sort_cap, int th) {
    FusedLds L;
    int o = 0;
    auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
    L.buf = take(2 * sort_cap * 8);
    L.spec = take((ntiles_img < SPEC_TILES ? ntiles_img : SPEC_TILES) * SPEC * 8);
    L.tcnt = take(ntiles_img * 4);
    L.toff = take((ntiles_img + 1) * 4);
    L.hist = take(2 * 512 * 4);
    L.flags = take(2 * (K > P ? K : P) * 4);
    {
        int np2k = 2;
        while (np2k < (K > P ? K : P)) np2k <<= 1;
        L.outk = out_keys(np2k);
        L.out = take(2 * L.outk * 8);
    }
    L.as_ = take(K * 4); L.posx = take(K * 4); L.posy = take(K * 4); L.ai_ = take(K * 4); L.ac_ = take(K * 4);
    L.ps_ = take(P * 4); L.pi_ = take(P * 4); L.pc_ = take(P * 4);
    const int nms = ((th + 2 * HALO) * LW + (th + 2 * HALO) * TW) * 4;
    L.total = o > nms ? o : nms;
    return L;
}

template <int TH_>
__global__ __launch_bounds__(FUSED_THREADS) void k_decode_fused(Group g0, Group g1, int h, int w, int tiles_x, int tiles, float min_score,
                                                                 uint64_t* cand, unsigned* records, int K, int P, int sort_cap,
                                                                 float conf, float dist_px, RegMaps rm, void* packed, int B) {
    constexpr int LH_ = TH_ + 2 * HALO, CAP_ = TW * TH_;      // tile height of this launch: fused_tile_height()
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int keep_n;
    __shared__ int misc[2][4];
    __shared__ int alive[2];
    __shared__ int wave_tot[FUSED_THREADS / 64];
    const int tid = threadIdx.x;
    const int C = g0.C + g1.C;
    const int nti = C * tiles;                                 // tiles of one image
    const int64_t blk = blockIdx.x;

    if (blk < (int64_t)B * nti) {
        // ================= tile block: clamped sigmoid + 5x5 NMS of one 64 x TH_ tile ==========================================
        float (*S)[LW] = reinterpret_cast<float(*)[LW]>(smem);
        float (*Hm)[TW] = reinterpret_cast<float(*)[TW]>(smem + sizeof(float) * LH_ * LW);
        const int b = (int)(blk / nti);
        const int rem = (int)(blk - (int64_t)b * nti);
        const int m = rem / tiles, tile = rem - m * tiles;     // map of the image: anchors 0..M-1, parts M..M+N-1
        const int grp = (m >= g0.C) ? 1 : 0;
        const Group g = grp ? g1 : g0;
        const int c = grp ? m - g0.C : m;
        const int tx0 = (tile % tiles_x) * TW;
        const int ty0 = (tile / tiles_x) * TH_;
        const float* plane = g.p + (int64_t)b * g.sb + (int64_t)c * g.sc;
        uint64_t* mine = cand + blk * CAP_;
        unsigned* rec = records + blk * REC_WORDS;
        SD_TRACE(blk < 1024 ? blk * 4 + 0 : -1);
        if (tid == 0) keep_n = 0;
        constexpr int NLD = (LH_ * LW + FUSED_THREADS - 1) / FUSED_THREADS;
        float ld[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * FUSED_THREADS;
            const int r = i / LW, cc = i - r * LW;
            const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
            const bool ok = i < LH_ * LW && y >= 0 && y < h && x >= 0 && x < w;
            ld[j] = plane[ok ? (int64_t)y * w + x : 0];
        }
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int i = tid + j * FUSED_THREADS;
            const int r = i / LW, cc = i - r * LW;
            const int y = ty0 + r - HALO, x = tx0 + cc - HALO;
            const bool ok = y >= 0 && y < h && x >= 0 && x < w;
            if (i < LH_ * LW) S[r][cc] = ok ? clamped_sigmoid(ld[j]) : -INFINITY;
        }
        __syncthreads();
        SD_TRACE(blk < 1024 ? blk * 4 + 1 : -1);
        for (int i = tid; i < LH_ * TW; i += FUSED_THREADS) {
            const int r = i / TW, cc = i - r * TW;
            float mx = fmaxf(fmaxf(S[r][cc], S[r][cc + 1]), fmaxf(S[r][cc + 2], S[r][cc + 3]));
            Hm[r][cc] = fmaxf(mx, S[r][cc + 4]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (TW * TH_) / FUSED_THREADS; ++j) {
            const int i = tid + j * FUSED_THREADS;
            const int r = i / TW, cc = i - r * TW;
            const int y = ty0 + r, x = tx0 + cc;
            float mx = fmaxf(fmaxf(Hm[r][cc], Hm[r + 1][cc]), fmaxf(Hm[r + 2][cc], Hm[r + 3][cc]));
            mx = fmaxf(mx, Hm[r + 4][cc]);
            const float v = S[r + HALO][cc + HALO];
            if ((y < h) && (x < w) && (v == mx) && (v >= min_score)) {        // `>=`: see k_nms_tile
                const int slot = atomicAdd(&keep_n, 1);
                const uint64_t key = make_key(v, (uint32_t)(c * h * w + y * w + x));
                __hip_atomic_store(mine + slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < SPEC)
                    __hip_atomic_store(reinterpret_cast<uint64_t*>(rec + 2) + slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // publish: every wave drains its write-through stores, then ONE lane sets the record's count word
        SD_TRACE(blk < 1024 ? blk * 4 + 2 : -1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(rec, REC_VALID | (unsigned)keep_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        SD_TRACE(blk < 1024 ? blk * 4 + 3 : -1);
        return;
    }

    // ================= selector block of image b: wait for the image's records, exact top-K / top-P, association ============
    const int b = (int)(blk - (int64_t)B * nti);
    const FusedLds Lo = fused_lds(K, P, nti, sort_cap, TH_);
    uint64_t* buf = reinterpret_cast<uint64_t*>(smem + Lo.buf);
    uint64_t* spec = reinterpret_cast<uint64_t*>(smem + Lo.spec);
    int* tcnt = reinterpret_cast<int*>(smem + Lo.tcnt);
    int* toff = reinterpret_cast<int*>(smem + Lo.toff);
    int* hist = reinterpret_cast<int*>(smem + Lo.hist);
    int* flags = reinterpret_cast<int*>(smem + Lo.flags);
    float* as_ = reinterpret_cast<float*>(smem + Lo.as_);
    float* ps_ = reinterpret_cast<float*>(smem + Lo.ps_);
    float* posx = reinterpret_cast<float*>(smem + Lo.posx);
    float* posy = reinterpret_cast<float*>(smem + Lo.posy);
    int* ai_ = reinterpret_cast<int*>(smem + Lo.ai_);
    int* ac_ = reinterpret_cast<int*>(smem + Lo.ac_);
    int* pi_ = reinterpret_cast<int*>(smem + Lo.pi_);
    int* pc_ = reinterpret_cast<int*>(smem + Lo.pc_);
    unsigned* rec_img = records + (int64_t)b * nti * REC_WORDS;
    const uint64_t* cand_img = cand + (int64_t)b * nti * CAP_;
    const PackedLayout L = packed_layout(packed, B, K, P);
    SD_TRACE(4096 + b * 8 + 0);
    if (tid < 2) alive[tid] = 0;
    {
        // thread t owns tiles t, t + 256, ...; `pending` = owned tiles whose record has not been taken yet
        int pending = 0;
        for (int t = tid; t < nti; t += FUSED_THREADS) ++pending;
        unsigned taken = 0;                                    // bit i: owned tile number i taken (nti <= 32 * 256 checked by the host)
        int polls = 0;
        while (true) {
            int i = 0;
            for (int t = tid; t < nti; t += FUSED_THREADS, ++i) {
                if (taken & (1u << i)) continue;
                unsigned* rec = rec_img + (int64_t)t * REC_WORDS;
                const unsigned word = __hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint64_t k6[SPEC];
#pragma unroll
                for (int u = 0; u < SPEC; ++u) k6[u] = ldkey<true>(reinterpret_cast<uint64_t*>(rec + 2) + u);
                if (!(word & REC_VALID)) continue;
                const int cnt = (int)(word & ~REC_VALID);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < SPEC; ++u) ok = ok && (u >= cnt || k6[u] != 0ull);
                if (!ok) continue;
                tcnt[t] = cnt;
                if (t < SPEC_TILES) {
#pragma unroll
                    for (int u = 0; u < SPEC; ++u) spec[t * SPEC + u] = k6[u];
                }
                __hip_atomic_store(rec, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            // state left zero
#pragma unroll
                for (int u = 0; u < SPEC; ++u)
                    if (u < cnt) __hip_atomic_store(reinterpret_cast<uint64_t*>(rec + 2) + u, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                taken |= 1u << i;
                --pending;
            }
            if (__syncthreads_or(pending) == 0) break;
            if (++polls > POLL_LIMIT) {                         // never hang the GPU: report and leave (block-uniform decision)
                if (tid == 0) L.status[b] = 1;
                return;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    SD_TRACE(4096 + b * 8 + 1);
    // exclusive prefix of the tile counts over the whole image (anchor tiles first): thread t owns a contiguous chunk
    {
        const int chunk = (nti + FUSED_THREADS - 1) / FUSED_THREADS;
        const int lo = min(tid * chunk, nti), hi = min(lo + chunk, nti);
        int sum = 0;
        for (int t = lo; t < hi; ++t) sum += tcnt[t];
        int incl = sum;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wave; ++q) before += wave_tot[q];
        int run = before + incl - sum;
        for (int t = lo; t < hi; ++t) { toff[t] = run; run += tcnt[t]; }
        if (tid == FUSED_THREADS - 1) toff[nti] = run;
    }
    __syncthreads();
    const int split = g0.C * tiles;                            // first part tile
    const int n0 = toff[split], n1 = toff[nti] - n0;
    __syncthreads();
    for (int t = split + tid; t < nti; t += FUSED_THREADS) toff[t] -= n0;   // part offsets relative to the part list
    const int hw = h * w;
    SD_TRACE(4096 + b * 8 + 2);
    {
        // anchors on waves 0-1, parts on waves 2-3; both teams run the same barrier sequence (path and sort size from the longer list)
        const int team = tid / FUSED_TEAM, ttid = tid - team * FUSED_TEAM;
        const int maxkp = K > P ? K : P;
        const Team T{ttid, buf + team * sort_cap, hist + team * 512, misc[team], flags + team * maxkp,
                     reinterpret_cast<uint64_t*>(smem + Lo.out) + team * Lo.outk, team, alive};
        const int n = team ? n1 : n0, k = team ? P : K;
        const int t0 = team ? split : 0;
        const TiledSrc<FUSED_TEAM> src{cand_img + (int64_t)t0 * CAP_, tcnt + t0, toff + t0, spec, t0, team ? nti - split : split, n, CAP_};
        team_select_topk<FUSED_TEAM>(T, src, k, max(n0, n1), maxkp, sort_cap);
        SD_TRACE(4096 + b * 8 + 3);
        fill_zero_slots<FUSED_TEAM>(T, min(n, k), k);
        SD_TRACE(4096 + b * 8 + 4);
        float* os = team ? ps_ : as_;
        int* oi = team ? pi_ : ai_;
        int* oc = team ? pc_ : ac_;
        for (int i = ttid; i < k; i += FUSED_TEAM) {
            const uint64_t key = T.buf[i];
            const uint32_t flat = ~(uint32_t)key;
            const int cls = flat / hw;
            os[i] = ord2f((uint32_t)(key >> 32)); oi[i] = flat - cls * hw; oc[i] = cls;
        }
    }
    __syncthreads();
    block_group(b, K, P, w, conf, dist_px, rm, as_, ai_, ac_, ps_, pi_, pc_, posx, posy, L);
    __syncthreads();
    SD_TRACE(4096 + b * 8 + 7);
}

// association from externally supplied peaks (sd_decode_group)
__global__ __launch_bounds__(SEL_THREADS) void k_group_only(const float* a_score, const int64_t* a_ind, const float* a_cls,
                                                             const float* p_score, const int64_t* p_ind, const float* p_cls,
                                                             int hw, int w, int K, int P, float conf, float dist_px, RegMaps rm,
                                                             void* packed, int B) {
    __shared__ float as_[SD_MAX_TOPK], ps_[SD_MAX_TOPK], posx[SD_MAX_TOPK], posy[SD_MAX_TOPK];
    __shared__ int ai_[SD_MAX_TOPK], ac_[SD_MAX_TOPK], pi_[SD_MAX_TOPK], pc_[SD_MAX_TOPK];
    const int b = blockIdx.x, tid = threadIdx.x;
    // indices come from the caller: clamp so that a bad index can never fault the GPU
    for (int i = tid; i < K; i += SEL_THREADS) {
        const int64_t id = a_ind[(int64_t)b * K + i];
        as_[i] = a_score[(int64_t)b * K + i]; ai_[i] = (int)(id < 0 ? 0 : (id >= hw ? hw - 1 : id)); ac_[i] = (int)a_cls[(int64_t)b * K + i];
    }
    for (int i = tid; i < P; i += SEL_THREADS) {
        const int64_t id = p_ind[(int64_t)b * P + i];
        ps_[i] = p_score[(int64_t)b * P + i]; pi_[i] = (int)(id < 0 ? 0 : (id >= hw ? hw - 1 : id)); pc_[i] = (int)p_cls[(int64_t)b * P + i];
    }
    __syncthreads();
    const PackedLayout L = packed_layout(packed, B, K, P);
    block_group(b, K, P, w, conf, dist_px, rm, as_, ai_, ac_, ps_, pi_, pc_, posx, posy, L);
}

// ---------------------------------------------------------------------------------------------
// small elementwise prims
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clamped_sigmoid(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = clamped_sigmoid(v.x); v.y = clamped_sigmoid(v.y); v.z = clamped_sigmoid(v.z); v.w = clamped_sigmoid(v.w);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) y[i] = clamped_sigmoid(x[i]);
}

__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ feat, int64_t sb, int64_t sc, int C, int64_t hw,
                                                 const int64_t* __restrict__ ind, int n, float* __restrict__ out, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * n * C) return;
    const int c = (int)(i % C);
    const int64_t bn = i / C;
    const int b = (int)(bn / n);
    int64_t id = ind[bn];
    id = id < 0 ? 0 : (id >= hw ? hw - 1 : id);   // the reference raises on out-of-range; never fault here
    out[i] = feat[(int64_t)b * sb + (int64_t)c * sc + id];
}

__global__ __launch_bounds__(256) void k_hypot(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = reinterpret_cast<const float2*>(in)[i];
    const float sx = v.x * v.x, sy = v.y * v.y;
    out[i] = sqrtf(sx + sy);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int check_map(const char* what, const void* p, int64_t sb, int64_t sc, int B, int C, int h, int w) {
    SD_REQUIRE(p != nullptr, SD_ERR_INVALID, "%s: null pointer", what);
    SD_REQUIRE(B > 0 && C > 0 && h > 0 && w > 0, SD_ERR_INVALID, "%s: bad shape (%d,%d,%d,%d)", what, B, C, h, w);
    SD_REQUIRE((int64_t)C * h * w < (1ll << 31), SD_ERR_INVALID, "%s: C*h*w must be < 2^31", what);
    SD_REQUIRE(sc >= (int64_t)h * w && (B == 1 || sb >= (int64_t)h * w), SD_ERR_INVALID, "%s: bad strides sb=%lld sc=%lld", what,
               (long long)sb, (long long)sc);
    return 0;
}

struct PeaksWs {
    int* counters;        // B*2 counters, CNT_STRIDE ints apart
    uint64_t* cand0;      // B*C0*h*w
    uint64_t* cand1;      // B*C1*h*w
    size_t bytes;
};
static PeaksWs carve(void* ws, int B, int C0, int C1, int h, int w) {
    PeaksWs r;
    const uintptr_t p = reinterpret_cast<uintptr_t>(ws);   // integer arithmetic: the size queries carve a null base (pointer + offset on null is UB)
    size_t off = 0;
    r.counters = reinterpret_cast<int*>(p + off);      off += align_up((size_t)B * 2 * CNT_STRIDE * sizeof(int), 256);
    r.cand0 = reinterpret_cast<uint64_t*>(p + off);    off += align_up((size_t)B * C0 * h * w * 8, 256);
    r.cand1 = reinterpret_cast<uint64_t*>(p + off);    off += align_up((size_t)B * C1 * h * w * 8, 256);
    r.bytes = off;
    return r;
}

}  // namespace sd

using namespace sd;

extern "C" {

int sd_clamped_sigmoid(const float* x, float* y, int64_t n, sd_stream_t stream) {
    SD_REQUIRE(x && y && n >= 0, SD_ERR_INVALID, "sd_clamped_sigmoid: bad arguments");
    SD_REQUIRE(aligned16(x) && aligned16(y), SD_ERR_ALIGN, "sd_clamped_sigmoid: pointers must be 16-byte aligned");
    if (n == 0) return 0;
    const int grid = (int)std::min<int64_t>(cdiv(n, 1024), 2048);
    hipLaunchKernelGGL(k_clamped_sigmoid, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_nms5(const float* hm, int64_t sb, int64_t sc, float* out, int B, int C, int h, int w, int apply_sigmoid,
            sd_stream_t stream) {
    if (int e = check_map("sd_nms5", hm, sb, sc, B, C, h, w)) return e;
    SD_REQUIRE(out != nullptr, SD_ERR_INVALID, "sd_nms5: null output");
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    Group g0{hm, sb, sc, C}, g1{nullptr, 0, 0, 0};
    hipLaunchKernelGGL(k_nms_tile<0>, dim3(tiles_x * tiles_y, C, B), dim3(256), 0, (hipStream_t)stream, g0, g1, h, w, tiles_x,
                       apply_sigmoid, 0.f, out, (uint64_t*)nullptr, (uint64_t*)nullptr, (int*)nullptr);
    SD_LAUNCH_CHECK();
    return 0;
}

size_t sd_topk_workspace_bytes(int B, int C, int h, int w, int k) {
    (void)k;
    return carve(nullptr, B, C, 0, h, w).bytes;
}

int sd_topk(const float* scores, int64_t sb, int64_t sc, int B, int C, int h, int w, int k, float* out_score, int64_t* out_ind,
            float* out_cls, float* out_ys, float* out_xs, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_map("sd_topk", scores, sb, sc, B, C, h, w)) return e;
    SD_REQUIRE(k > 0 && k <= SD_MAX_TOPK && (int64_t)k <= (int64_t)C * h * w, SD_ERR_INVALID, "sd_topk: k=%d out of range", k);
    SD_REQUIRE(out_score && out_ind && out_cls && out_ys && out_xs && workspace, SD_ERR_INVALID, "sd_topk: null pointer");
    const PeaksWs ws = carve(workspace, B, C, 0, h, w);
    SD_REQUIRE(workspace_bytes >= ws.bytes, SD_ERR_WORKSPACE, "sd_topk: workspace %zu < %zu", workspace_bytes, ws.bytes);
    const int hw = h * w;
    hipLaunchKernelGGL(k_dense_keys, dim3(cdiv(hw, 256), C, B), dim3(256), 0, (hipStream_t)stream, scores, sb, sc, C, hw, ws.cand0);
    SD_LAUNCH_CHECK();
    PeakOut out{out_score, out_ind, out_cls, out_ys, out_xs};
    hipLaunchKernelGGL(k_select_peaks, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, ws.cand0, (const int*)nullptr, 0,
                       (int64_t)C * hw, C * hw, k, hw, w, 0, out);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_transpose_and_gather(const float* feat, int64_t sb, int64_t sc, int B, int C, int64_t hw, const int64_t* ind, int n,
                            float* out, sd_stream_t stream) {
    SD_REQUIRE(feat && ind && out && B > 0 && C > 0 && hw > 0 && n >= 0, SD_ERR_INVALID, "sd_transpose_and_gather: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gather, dim3(cdiv((int64_t)B * n * C, 256)), dim3(256), 0, (hipStream_t)stream, feat, sb, sc, C, hw, ind, n,
                       out, B);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_hypot(const float* in_pairs, float* out, int64_t n, sd_stream_t stream) {
    SD_REQUIRE(in_pairs && out && n >= 0, SD_ERR_INVALID, "sd_hypot: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_hypot, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, in_pairs, out, n);
    SD_LAUNCH_CHECK();
    return 0;
}

size_t sd_decode_peaks_workspace_bytes(int B, int C, int h, int w, int k) {
    (void)k;
    return carve(nullptr, B, C, 0, h, w).bytes;
}

int sd_decode_peaks(const float* logits, int64_t sb, int64_t sc, int B, int C, int h, int w, int k, float* out_score,
                    int64_t* out_ind, float* out_cls, float* out_ys, float* out_xs, void* workspace, size_t workspace_bytes,
                    sd_stream_t stream) {
    if (int e = check_map("sd_decode_peaks", logits, sb, sc, B, C, h, w)) return e;
    SD_REQUIRE(k > 0 && k <= SD_MAX_TOPK && (int64_t)k <= (int64_t)C * h * w, SD_ERR_INVALID, "sd_decode_peaks: k=%d out of range", k);
    SD_REQUIRE(out_score && out_ind && out_cls && out_ys && out_xs && workspace, SD_ERR_INVALID, "sd_decode_peaks: null pointer");
    const PeaksWs ws = carve(workspace, B, C, 0, h, w);
    SD_REQUIRE(workspace_bytes >= ws.bytes, SD_ERR_WORKSPACE, "sd_decode_peaks: workspace %zu < %zu", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    SD_HIP(hipMemsetAsync(ws.counters, 0, (size_t)B * 2 * CNT_STRIDE * sizeof(int), st));
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    Group g0{logits, sb, sc, C}, g1{nullptr, 0, 0, 0};
    hipLaunchKernelGGL(k_nms_tile<1>, dim3(tiles_x * tiles_y, C, B), dim3(256), 0, st, g0, g1, h, w, tiles_x, 1, 0.f, (float*)nullptr,
                       ws.cand0, ws.cand1, ws.counters);
    SD_LAUNCH_CHECK();
    PeakOut out{out_score, out_ind, out_cls, out_ys, out_xs};
    hipLaunchKernelGGL(k_select_peaks, dim3(B), dim3(SEL_THREADS), 0, st, ws.cand0, ws.counters, 2 * CNT_STRIDE, (int64_t)C * h * w, 0, k, h * w,
                       w, 1, out);
    SD_LAUNCH_CHECK();
    return 0;
}

// ---- map-parallel path of sd_decode (k_nms_slots -> k_select_map -> k_rank_maps -> k_group_wide) ----
struct MapWs {
    uint64_t* cand;       // B * C * tiles * (TW * th) keys
    int* tile_cnt;        // B * C * tiles
    uint64_t* stage1;     // B * C * max(K, P) keys
    int* stage1_cnt;      // B * C
    uint64_t* final_keys; // B * 2 * max(K, P) keys
    int* final_cnt;       // B * 2
    size_t bytes;
};
constexpr int MAP_SPLIT_MAX = 4;            // parts a map is split into at most (k_map_stream_select)
static MapWs carve_map(void* ws, int B, int C, int h, int w, int th, int K, int P) {
    MapWs r;
    const uintptr_t p = reinterpret_cast<uintptr_t>(ws);   // integer arithmetic: size queries carve a null base
    const size_t tiles = (size_t)cdiv(w, TW) * cdiv(h, th);
    size_t off = 0;
    r.cand = reinterpret_cast<uint64_t*>(p + off);     off += align_up((size_t)B * C * tiles * TW * th * 8, 256);
    r.tile_cnt = reinterpret_cast<int*>(p + off);      off += align_up((size_t)B * C * tiles * sizeof(int), 256);
    r.stage1 = reinterpret_cast<uint64_t*>(p + off);   off += align_up((size_t)B * C * MAP_SPLIT_MAX * std::max(K, P) * 8, 256);   // one list per part of a split map
    r.stage1_cnt = reinterpret_cast<int*>(p + off);    off += align_up((size_t)B * C * sizeof(int), 256);
    r.final_keys = reinterpret_cast<uint64_t*>(p + off); off += align_up((size_t)B * 2 * std::max(K, P) * 8, 256);
    r.final_cnt = reinterpret_cast<int*>(p + off);     off += align_up((size_t)B * 2 * sizeof(int), 256);
    r.bytes = off;
    return r;
}
// Tile blocks (at 64x16) from which sd_decode takes the map-parallel path, and its tile height (0 = by size); per host thread.
// sd_decode_set_option("map_parallel_from" / "map_tile_height", n).
static thread_local int g_map_parallel_from = 2560;
static thread_local int g_map_from_user = 0;        // 1: "map_parallel_from" was set by the caller and holds for every geometry and mode (tests, A/B)
static thread_local int g_map_tile_height = 0;
static thread_local int g_map_stream = 1;           // 0: tile kernel + k_select_map instead of k_map_stream_select (A/B, tests)
static thread_local int g_map_rows11 = 1;           // bands of maps with 9-16 units of 11 rows: 1 = 8 or 11 rows by size, 0 = 16 rows, 8 / 11 = forced (A/B, tests)
static thread_local int g_map_scalar_nms = 0;       // 1: the per-pixel-sigmoid tile kernel also where the logit-domain one applies (A/B, tests)
static thread_local int g_map_waves3 = 1;           // parts of three wave-iterations on 192-thread blocks: 1 = from 1024 blocks with a score threshold, 0 never, 2 always (A/B, tests)
static thread_local int g_map_half = 1;             // 0: one band per wave also on maps up to 128 columns wide (A/B, tests)
// 64 x 16 tile blocks per call from which sd_decode takes the map-parallel path (and the one-launch kernel is no longer recommended).  On maps
// up to 128 columns wide -- two bands per wave, two parts per map, one rank + association launch -- measured at the cfg shape after the early
// score cut (`profiles/r05_decode_small_batches.txt`): annotations-only 16.7 us at bs = 16 .. 20 against 16.6 .. 17.7 for k_decode_fused and
// 19.8 for the launch pair (bs = 32: 15.7 / 19.2; bs = 8: 16.3 / 15.9 / 20.3): from 960 tile blocks (bs = 20); with the exact top-k 19.7 us
// at bs = 1 against 26.5 for k_decode_fused: always.  Wider maps keep the round-4 threshold.
static int64_t map_from(int w, bool exact) {
    if (g_map_from_user) return g_map_parallel_from;
    const bool fast = g_map_half && g_map_stream && w <= 128 && w % 4 == 0;
    return fast ? (exact ? 1 : 960) : g_map_parallel_from;
}
// ... and inside sd_decode, where the alternative is the launch pair (19.8-21.6 us at bs = 1 .. 16 against 14.8-16.8): always on those maps
static int64_t map_from_pair(int w) {
    if (g_map_from_user) return g_map_parallel_from;
    return (g_map_half && g_map_stream && w <= 128 && w % 4 == 0) ? 1 : g_map_parallel_from;
}
static thread_local int g_map_split = 0;            // parts per map in k_map_stream_select: 0 = by geometry, 1 .. MAP_SPLIT_MAX = forced (A/B, tests)
static thread_local int g_map_rank_group = 1;       // 0: k_rank_maps + k_group_wide instead of the one-launch k_rank_group (A/B, tests)
constexpr size_t RANK_GROUP_LDS_MAX = 96 * 1024;    // dynamic LDS of k_rank_group (beside its 37 KB of static arrays)
static int map_tile_height(int64_t blocks16) {
    if (g_map_tile_height == 16 || g_map_tile_height == 32) return g_map_tile_height;
    return blocks16 >= 8192 ? 32 : 16;
}
static bool map_path_possible(int M, int N, int h, int w, int K, int P) {
    return M <= 64 && N <= 64 && (int64_t)cdiv(w, TW) * cdiv(h, 16) <= MAP_TILES_MAX && K <= SD_MAX_TOPK && P <= SD_MAX_TOPK &&
           (int64_t)M * K <= RANK_KEYS_MAX && (int64_t)N * P <= RANK_KEYS_MAX;
}

size_t sd_decode_workspace_bytes(int B, int M, int N, int h, int w, int K, int P) {
    size_t need = carve(nullptr, B, M, N, h, w).bytes;
    if (map_path_possible(M, N, h, w, K, P))           // either tile height (the option may change between sizing and launch)
        need = std::max({need, carve_map(nullptr, B, M + N, h, w, 16, K, P).bytes, carve_map(nullptr, B, M + N, h, w, 32, K, P).bytes});
    return need;
}

size_t sd_decode_packed_words(int B, int K, int P) { return (size_t)B * (6 * (size_t)K + 11 * (size_t)P + 1); }

int sd_decode(const float* anchor_hm, int64_t a_sb, int64_t a_sc, const float* part_hm, int64_t p_sb, int64_t p_sc,
              const float* offsets, int64_t o_sb, int64_t o_sc, const float* embeddings, int64_t e_sb, int64_t e_sc, int B, int M,
              int N, int h, int w, int K, int P, float conf, float dist_px, int exact_topk, void* packed, void* workspace,
              size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_map("sd_decode(anchor_hm)", anchor_hm, a_sb, a_sc, B, M, h, w)) return e;
    if (int e = check_map("sd_decode(part_hm)", part_hm, p_sb, p_sc, B, N, h, w)) return e;
    if (int e = check_map("sd_decode(offsets)", offsets, o_sb, o_sc, B, 2, h, w)) return e;
    if (int e = check_map("sd_decode(embeddings)", embeddings, e_sb, e_sc, B, 2, h, w)) return e;
    SD_REQUIRE(K > 0 && K <= SD_MAX_TOPK && (int64_t)K <= (int64_t)M * h * w, SD_ERR_INVALID, "sd_decode: max_objects=%d out of range", K);
    SD_REQUIRE(P > 0 && P <= SD_MAX_TOPK && (int64_t)P <= (int64_t)N * h * w, SD_ERR_INVALID, "sd_decode: max_parts=%d out of range", P);
    SD_REQUIRE(packed && workspace, SD_ERR_INVALID, "sd_decode: null pointer");
    hipStream_t st = (hipStream_t)stream;
    Group g0{anchor_hm, a_sb, a_sc, M}, g1{part_hm, p_sb, p_sc, N};
    const int64_t blocks16 = (int64_t)B * (M + N) * cdiv(w, TW) * cdiv(h, 16);
    // measured (tools/decode_path_sweep.py): the launch pair costs ~20 us + what ONE block per image needs for its lists (1024x1024, 8 + 8
    // maps: 64 us at bs = 1), the map-parallel chain 25-35 us + the tile pass; they cross at ~2500 tile blocks, and images of 1024 and
    // more tile blocks are better off on the map-parallel path at any batch size
    const int64_t tiles_img16 = (int64_t)(M + N) * cdiv(w, TW) * cdiv(h, 16);
    const bool want_map = blocks16 >= map_from_pair(w) || (g_map_parallel_from < (1 << 30) && tiles_img16 >= 1024);
    if (map_path_possible(M, N, h, w, K, P) && want_map && blocks16 < (1ll << 30)) {
        const int th = map_tile_height(blocks16);
        const MapWs mw = carve_map(workspace, B, M + N, h, w, th, K, P);
        SD_REQUIRE(workspace_bytes >= mw.bytes, SD_ERR_WORKSPACE, "sd_decode: workspace %zu < %zu", workspace_bytes, mw.bytes);
        const int tiles_x = cdiv(w, TW), tiles = tiles_x * cdiv(h, th), C = M + N;
        const float min_score = exact_topk ? 0.f : conf;
        const bool vec = g_map_scalar_nms == 0 && w % 4 == 0 && aligned16(anchor_hm) && aligned16(part_hm) && a_sb % 4 == 0 && a_sc % 4 == 0 &&
                         p_sb % 4 == 0 && p_sc % 4 == 0;
        int splits = 1;                                              // stage-1 lists per map (parts of a split map)
        if (vec && g_map_scalar_nms == 0 && g_map_stream) {
            // tile pass + per-map selection in one kernel (candidate list `cand`: h * w slots per map, only touched by overflowing maps)
            // 16 waves per map where a map has 16+ units of work (strip x 16-row band): 256 x 256 maps 40.9 -> 36.4 us per batch of 16 x 16 maps;
            // 128 x 128 maps (8 units) stay at 8 waves (14.4 us; 14.9 with 16)
            const int strips = cdiv(w, 256);
            // 16 waves per map where 16-row bands give 16+ units of work (256 x 256 maps: 40.9 -> 36.5 us per batch of 16 x 16 maps); maps of
            // 9 .. 15 such units (128 x 128: 8) take 11-row bands: every wave walks 15 rows instead of 20 -- and, round 5, such a map is SPLIT
            // over ceil(bands / 4) blocks of four waves, one band per wave (bs = 64, 3 maps of 128 x 128: 576 blocks of 4 waves on every CU of
            // the chip instead of 192 blocks with 12 busy waves of 16)
            const int units16 = strips * cdiv(h, 16), units11 = strips * cdiv(h, 11);
            // ... and 8-row bands (12 rows per wave, all requested at once) with two bands per wave while the launch is small enough for its
            // latency to matter more than the halo rows it re-reads (measured at 128 x 128 maps, two parts per map: bs = 64 17.3 -> 16.6 us,
            // bs = 128 19.2 -> 18.2, exact top-k 24.6 -> 23.3; bs = 512 36.5 -> 38.1: from 1024 maps the 11-row bands stay)
            const bool half_ok = g_map_half && strips == 1 && w <= 128;
            const bool short_ok = units11 > 8 && units11 <= 16;
            const int rows = units16 >= 16 ? 16 : (!short_ok || g_map_rows11 == 0) ? 16
                           : (g_map_rows11 == 8 || (g_map_rows11 == 1 && half_ok && (int64_t)B * C < 1024)) ? (half_ok ? 8 : 11) : 11;
            const int bands = cdiv(h, rows);
            // maps up to 128 columns wide: two bands per wave (lanes 0-31 / 32-63), half the waves (sd_decode_set_option("map_half", 0): off)
            const bool half = g_map_half && strips == 1 && w <= 128;
            int want = g_map_split;
            // (measured, `profiles/r05_decode_split_sweep.txt`: at bs = 64 -- 192 maps -- one block per map and three parts per map stream in the
            // same 14 us, the kernel is bound by the start-up spread of its ~2500 waves and their first round trips, and the rank kernel pays for
            // three times the lists; at bs = 512 the parts stream in 53 instead of 74 us: five 27 KB blocks per CU instead of one of 86 KB)
            //  With two bands per wave (maps up to 128 columns) two parts per map pay from ~100 maps: bs = 64 21.4 -> 20.5 us, bs = 512 72.8 -> 53.6.)
            if (want <= 0) want = ((rows == 11 || rows == 8) && strips == 1) ? (half ? ((int64_t)B * C >= 96 ? 2 : 1) : ((int64_t)B * C >= 384 ? cdiv(bands, 4) : 1)) : 1;
            splits = std::max(1, std::min({want, MAP_SPLIT_MAX, bands}));
            if ((int64_t)M * splits * K > RANK_KEYS_MAX || (int64_t)N * splits * P > RANK_KEYS_MAX) splits = 1;
            const int per_block = strips * cdiv(bands, splits);            // units of work of the largest part
            const int per_wave = half ? cdiv(per_block, 2) : per_block;    // wave-iterations of work of the largest part
            const unsigned grid = (unsigned)(B * C * splits);
            const float min_logit = conservative_min_logit(min_score);
            const bool waves3 = g_map_waves3 == 2 || (g_map_waves3 == 1 && grid >= 1024u && min_score > 0.f);
#define SD_STREAM(NT_, ROWS_, HALF_) hipLaunchKernelGGL((k_map_stream_select<NT_, ROWS_, HALF_>), dim3(grid), dim3(NT_), 0, st, g0, g1, h, w, min_score, min_logit, K, P, mw.cand, mw.stage1, splits)
            if (half) {
                if (rows == 8) { if (per_wave > 8) SD_STREAM(1024, 8, true); else if (per_wave > 4) SD_STREAM(512, 8, true); else SD_STREAM(256, 8, true); }
                else if (rows == 16) { if (per_wave > 8) SD_STREAM(1024, 16, true); else if (per_wave > 4) SD_STREAM(512, 16, true); else SD_STREAM(256, 16, true); }
                // (three band pairs per part -- the cfg shape in two parts -- on three waves where several blocks share a CU and a score
                //  threshold keeps the selections short: bs = 512 32.1 -> 28.7 us; at bs = 64, 384 blocks, the idle fourth wave costs nothing
                //  -- 9.6 vs 10.1 us: the start-up spread of a launch goes by its blocks, not its waves -- and the exact top-k's radix select over
                //  ~1000 keys wants the fourth wave: 16.3 vs 21.0 us)
                else            { if (per_wave > 8) SD_STREAM(1024, 11, true); else if (per_wave > 4) SD_STREAM(512, 11, true); else if (per_wave == 3 && waves3) SD_STREAM(192, 11, true); else SD_STREAM(256, 11, true); }
            } else if (rows == 16) {
                if (splits == 1 && units16 >= 16) SD_STREAM(1024, 16, false);
                else if (per_block > 4 || splits == 1) SD_STREAM(512, 16, false);
                else SD_STREAM(256, 16, false);
            } else {
                if (per_block > 8 || splits == 1) SD_STREAM(1024, 11, false);
                else if (per_block > 4) SD_STREAM(512, 11, false);
                else if (per_block == 3 && waves3) SD_STREAM(192, 11, false);
                else SD_STREAM(256, 11, false);
            }
#undef SD_STREAM
            SD_LAUNCH_CHECK();
        } else {
            const dim3 tgrid((unsigned)((int64_t)B * C * tiles));
            if (vec && th == 32) hipLaunchKernelGGL(k_nms_slots_v<32>, tgrid, dim3(256), 0, st, g0, g1, h, w, tiles_x, tiles, min_score, mw.cand, mw.tile_cnt);
            else if (vec)        hipLaunchKernelGGL(k_nms_slots_v<16>, tgrid, dim3(256), 0, st, g0, g1, h, w, tiles_x, tiles, min_score, mw.cand, mw.tile_cnt);
            else if (th == 32)   hipLaunchKernelGGL(k_nms_slots<32>, tgrid, dim3(256), 0, st, g0, g1, h, w, tiles_x, tiles, min_score, mw.cand, mw.tile_cnt);
            else                 hipLaunchKernelGGL(k_nms_slots<16>, tgrid, dim3(256), 0, st, g0, g1, h, w, tiles_x, tiles, min_score, mw.cand, mw.tile_cnt);
            SD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_select_map, dim3(B * C), dim3(SEL_THREADS), 0, st, mw.cand, mw.tile_cnt, tiles, TW * th, M, N, K, P, mw.stage1);
            SD_LAUNCH_CHECK();
        }
        RegMaps rm{offsets, o_sb, o_sc, embeddings, e_sb, e_sc};
        const int LM = M * splits, LN = N * splits;                  // lists per group
        const size_t rg_lds = ((size_t)LM * K + (size_t)LN * P + K + P) * 8;
        // ranks + gathers + association of an image in ONE block (two dependent launches less) where an image's lists are short: the
        // ranking is serial per image there (stress, 16 lists of 128 / 512 keys in one 1024-thread block: 75 us against 8.6 + 7.4 for
        // k_rank_maps + k_group_wide -- `profiles/r05_decode_split_sweep.txt`), so large selections keep the map-parallel pair.
        // map_rank_group: 0 never, 1 by size (default), 2 the generic one-block kernel wherever its lists fit LDS (tests)
        const int64_t n_keys = (int64_t)LM * K + (int64_t)LN * P;
        const bool small = K <= RGS_K && P <= RGS_K && LM * K <= RGS_KEYS && LN * P <= RGS_KEYS && n_keys <= RGS_THREADS * RGS_KPT && h * w >= RGS_K;
        if (g_map_rank_group == 1 && small) {
            hipLaunchKernelGGL(k_rank_group_small, dim3(B), dim3(RGS_THREADS), 0, st, mw.stage1, LM, LN, h, w, K, P, conf, dist_px, rm, packed, B);
            SD_LAUNCH_CHECK();
            return 0;
        }
        if (rg_lds <= RANK_GROUP_LDS_MAX && (g_map_rank_group == 2 || (g_map_rank_group == 1 && n_keys <= 1024 && K <= 256 && P <= 256))) {
            static thread_local bool raised_rg = false;             // per host thread: cheap, idempotent
            if (!raised_rg) {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_group<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RANK_GROUP_LDS_MAX));
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_group<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)RANK_GROUP_LDS_MAX));
                raised_rg = true;
            }
            if (n_keys <= 1024 && K <= 256 && P <= 256)
                hipLaunchKernelGGL(k_rank_group<256>, dim3(B), dim3(256), rg_lds, st, mw.stage1, LM, LN, h, w, K, P, conf, dist_px, rm, packed, B);
            else
                hipLaunchKernelGGL(k_rank_group<1024>, dim3(B), dim3(1024), rg_lds, st, mw.stage1, LM, LN, h, w, K, P, conf, dist_px, rm, packed, B);
            SD_LAUNCH_CHECK();
            return 0;
        }
        const size_t rank_lds = (size_t)std::max((int64_t)LM * K, (int64_t)LN * P) * 8;
        if (rank_lds > 48 * 1024) {
            static thread_local bool raised = false;               // per host thread: cheap, idempotent
            if (!raised) {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rank_maps), hipFuncAttributeMaxDynamicSharedMemorySize, RANK_KEYS_MAX * 8));
                raised = true;
            }
        }
        // (a part of a split map is one more sorted list of its group: k_rank_maps sees M * splits and N * splits lists)
        hipLaunchKernelGGL(k_rank_maps, dim3(B * C * splits), dim3(SEL_THREADS), rank_lds, st, mw.stage1, LM, LN, K, P, mw.final_keys);
        SD_LAUNCH_CHECK();
        const size_t group_lds = (size_t)K * 8 + (size_t)P * 8 + (size_t)((K + 3) & ~3) * 8 + (size_t)(K + P) * 4;
        hipLaunchKernelGGL(k_group_wide, dim3(B, cdiv(P, GROUP_PARTS)), dim3(GROUP_THREADS), group_lds, st, mw.final_keys, h, w, K, P,
                           conf, dist_px, rm, packed, B);
        SD_LAUNCH_CHECK();
        return 0;
    }
    const PeaksWs ws = carve(workspace, B, M, N, h, w);
    SD_REQUIRE(workspace_bytes >= ws.bytes, SD_ERR_WORKSPACE, "sd_decode: workspace %zu < %zu", workspace_bytes, ws.bytes);
    SD_HIP(hipMemsetAsync(ws.counters, 0, (size_t)B * 2 * CNT_STRIDE * sizeof(int), st));
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    hipLaunchKernelGGL(k_nms_tile<1>, dim3(tiles_x * tiles_y, M + N, B), dim3(256), 0, st, g0, g1, h, w, tiles_x, 1, exact_topk ? 0.f : conf, (float*)nullptr,
                       ws.cand0, ws.cand1, ws.counters);
    SD_LAUNCH_CHECK();
    RegMaps rm{offsets, o_sb, o_sc, embeddings, e_sb, e_sc};
    hipLaunchKernelGGL(k_select_group, dim3(B), dim3(2 * SEL_THREADS), 0, st, ws.cand0, ws.cand1, ws.counters, M, N, h, w, K, P, conf,
                       dist_px, rm, packed, B);
    SD_LAUNCH_CHECK();
    return 0;
}

constexpr int FUSED_LDS_LIMIT = 96 * 1024;    // dynamic LDS the launcher asks for at most (gfx950: 160 KB per CU)
static int next_pow2_host(int v) { int p = 1; while (p < v) p <<= 1; return p; }
// Tile height of one fused launch.  64x16 tiles give a small batch the most workgroups (bs=1: 32 + 1); once the launch holds more
// tile blocks than the chip keeps resident at once (about 1200 with the selector's LDS block), they run in rounds and 64x32 tiles --
// half the blocks, half the records the selectors wait for, 36 halo rows per 32 instead of 20 per 16 -- finish sooner.  Measured
// (tools/decode_tile_sweep.py, 3 maps of 128x128, us per launch 64x16 / 64x32): bs=1 12.8 / 14.0, bs=16 14.6 / 15.2, bs=32 16.6 / 16.7,
// bs=48 18.7 / 18.9, bs=64 21.1 / 20.1, bs=96 25.3 / 23.6, bs=128 29.6 / 27.1: the switch sits at 2688 blocks (bs=56).  sd_decode_set_option("tall_tiles_from", n) moves the switch.
static thread_local int g_tall_tiles_from = 2688;      // per host thread, like the conv dispatch thresholds
int sd_decode_set_option(const char* name, int value) {
    if (name && !strcmp(name, "tall_tiles_from")) { g_tall_tiles_from = value; return 0; }
    if (name && !strcmp(name, "map_parallel_from")) {            // < 0: back to the built-in rule (map_from)
        g_map_from_user = value >= 0; g_map_parallel_from = value >= 0 ? value : 2560; return 0;
    }
    if (name && !strcmp(name, "map_tile_height")) { g_map_tile_height = value; return 0; }
    if (name && !strcmp(name, "map_scalar_nms")) { g_map_scalar_nms = value; return 0; }
    if (name && !strcmp(name, "map_rows11")) { g_map_rows11 = value; return 0; }
    if (name && !strcmp(name, "map_stream")) { g_map_stream = value; return 0; }
    if (name && !strcmp(name, "map_split")) { g_map_split = value; return 0; }
    if (name && !strcmp(name, "map_half")) { g_map_half = value; return 0; }
    if (name && !strcmp(name, "map_waves3")) { g_map_waves3 = value; return 0; }
    if (name && !strcmp(name, "map_rank_group")) { g_map_rank_group = value; return 0; }
    sd::set_error("sd_decode_set_option: unknown option '%s'", name ? name : "(null)");
    return SD_ERR_INVALID;
}
static int fused_tile_height(int B, int M, int N, int h, int w) {
    const int64_t blocks16 = (int64_t)std::max(B, 1) * (M + N) * cdiv(w, TW) * cdiv(h, 16);
    return blocks16 >= g_tall_tiles_from ? 32 : 16;
}
static size_t fused_tiles(int h, int w, int th) { return (size_t)cdiv(w, TW) * cdiv(h, th); }

// both sizes cover EITHER tile height (the option may change between sizing and launch)
size_t sd_decode_state_bytes(int B, int M, int N, int h, int w) {
    return align_up((size_t)std::max(B, 1) * (M + N) * fused_tiles(h, w, 16) * REC_WORDS * sizeof(unsigned), 256);
}

size_t sd_decode_fused_workspace_bytes(int B, int M, int N, int h, int w, int K, int P) {
    (void)K; (void)P;
    const size_t t = std::max(fused_tiles(h, w, 16) * TW * 16, fused_tiles(h, w, 32) * TW * 32);
    return align_up((size_t)B * (M + N) * t * 8, 256);
}

#ifdef SD_DECODE_TRACE
int sd_debug_read_trace(unsigned long long* out, int n) {
    SD_HIP(hipDeviceSynchronize());
    SD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(sd::sd_trace), sizeof(unsigned long long) * (size_t)n));
    return 0;
}
#endif

int sd_selfcheck_sigmoid(unsigned long long* out3, sd_stream_t stream) {
    SD_REQUIRE(out3 != nullptr, SD_ERR_INVALID, "sd_selfcheck_sigmoid: null pointer");
    SD_HIP(hipMemsetAsync(out3, 0, 3 * sizeof(unsigned long long), (hipStream_t)stream));
    hipLaunchKernelGGL(k_selfcheck_sigmoid, dim3(4096), dim3(256), 0, (hipStream_t)stream, out3);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_stream_synchronize(sd_stream_t stream) {
    SD_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int sd_decode_fused_supported(int B, int M, int N, int h, int w, int K, int P) {
    if (B <= 0 || M <= 0 || N <= 0 || h <= 0 || w <= 0 || K <= 0 || P <= 0) return 0;
    // judged on the 64x16 tiling (more tiles, more LDS for their counts): what fits there fits with 64x32 tiles
    const int64_t nti = (int64_t)(M + N) * fused_tiles(h, w, 16);
    return K <= FUSED_MAX_TOPK && P <= FUSED_MAX_TOPK && B <= 256 && nti <= 32 * FUSED_THREADS && (int64_t)B * nti + B < (1ll << 31) &&
           fused_lds(K, P, (int)nti, std::max(FUSED_CAP_EXACT, 2 * next_pow2_host(std::max(K, P))), 32).total <= FUSED_LDS_LIMIT;
}

// Where ONE launch is the faster decoder (measured: profiles/r02_decode_variants.txt): image geometries of at most 256 tile blocks
// (512x512 with 2 + 1 maps = 48; the selector's LDS is paid by every tile block of the grid, and with ~1000 tiles per image -- 1024x1024,
// 8 + 8 maps -- it halves the occupancy of the 16 k tile blocks: 451 us vs 116 us for sd_decode at K = 128, P = 512), and for the exact
// top-k only small batches (bs = 64: 39.0 vs 31.3 us, its two 2048-key sort buffers cost the tile blocks occupancy).
static int64_t fused_image_tiles(int M, int N, int h, int w) { return (int64_t)(M + N) * fused_tiles(h, w, 16); }
int sd_decode_fused_recommended(int B, int M, int N, int h, int w, int K, int P, int exact_topk) {
    // round 5: from the batch where sd_decode takes its map-parallel path (2560 tile blocks per call: bs >= 54 at the cfg shape) that path is
    // the faster one on maps up to 128 columns wide -- two bands per wave, two parts per map, ranks + association in one launch: 20.5 us per
    // bs = 64 batch against 23.5 for the one-launch kernel (`profiles/r05_decode_split_sweep.txt`)
    const int64_t blocks16 = (int64_t)B * (M + N) * cdiv(w, TW) * cdiv(h, 16);
    if (blocks16 >= map_from(w, (exact_topk & 1) != 0) && g_map_half && g_map_stream && w <= 128 && w % 4 == 0 && map_path_possible(M, N, h, w, K, P)) return 0;
    return sd_decode_fused_supported(B, M, N, h, w, K, P) && fused_image_tiles(M, N, h, w) <= 256 && (!(exact_topk & 1) || B <= 8);
}

int sd_decode_fused(const float* anchor_hm, int64_t a_sb, int64_t a_sc, const float* part_hm, int64_t p_sb, int64_t p_sc,
                    const float* offsets, int64_t o_sb, int64_t o_sc, const float* embeddings, int64_t e_sb, int64_t e_sc, int B,
                    int M, int N, int h, int w, int K, int P, float conf, float dist_px, int exact_topk, void* packed, void* state,
                    size_t state_bytes, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_map("sd_decode_fused(anchor_hm)", anchor_hm, a_sb, a_sc, B, M, h, w)) return e;
    if (int e = check_map("sd_decode_fused(part_hm)", part_hm, p_sb, p_sc, B, N, h, w)) return e;
    if (int e = check_map("sd_decode_fused(offsets)", offsets, o_sb, o_sc, B, 2, h, w)) return e;
    if (int e = check_map("sd_decode_fused(embeddings)", embeddings, e_sb, e_sc, B, 2, h, w)) return e;
    SD_REQUIRE(K > 0 && K <= FUSED_MAX_TOPK && (int64_t)K <= (int64_t)M * h * w, SD_ERR_INVALID,
               "sd_decode_fused: max_objects=%d out of range (1..%d; use sd_decode beyond)", K, FUSED_MAX_TOPK);
    SD_REQUIRE(P > 0 && P <= FUSED_MAX_TOPK && (int64_t)P <= (int64_t)N * h * w, SD_ERR_INVALID,
               "sd_decode_fused: max_parts=%d out of range (1..%d; use sd_decode beyond)", P, FUSED_MAX_TOPK);
    // at most B selector blocks wait inside the grid: keep them far below the resident block slots of the chip (256 CUs x >= 2)
    SD_REQUIRE(B <= 256, SD_ERR_INVALID, "sd_decode_fused: batch %d > 256 (selector blocks must stay resident); use sd_decode", B);
    SD_REQUIRE(packed && workspace && state, SD_ERR_INVALID, "sd_decode_fused: null pointer");
    // exact_topk bit 1 (value 2): accept geometries where the two-launch sd_decode is the faster decoder (tests of this kernel)
    const bool force = (exact_topk & 2) != 0;
    exact_topk &= 1;
    SD_REQUIRE(force || (M > 0 && N > 0 && fused_image_tiles(M, N, h, w) <= 256), SD_ERR_INVALID,
               "sd_decode_fused: %lld tile blocks per image (> 256): sd_decode is several times faster for this geometry "
               "(sd_decode_fused_recommended() == 0); pass exact_topk | 2 to run it here anyway", (long long)fused_image_tiles(std::max(M, 1), std::max(N, 1), h, w));
    const int th = fused_tile_height(B, M, N, h, w);
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, th);
    const int tiles = tiles_x * tiles_y;
    const int64_t nti = (int64_t)(M + N) * tiles;
    SD_REQUIRE(nti <= 32 * FUSED_THREADS, SD_ERR_INVALID, "sd_decode_fused: %lld tiles per image > %d; use sd_decode", (long long)nti,
               32 * FUSED_THREADS);
    SD_REQUIRE((int64_t)B * nti + B < (1ll << 31), SD_ERR_INVALID, "sd_decode_fused: grid too large");
    SD_REQUIRE(state_bytes >= sd_decode_state_bytes(B, M, N, h, w), SD_ERR_WORKSPACE, "sd_decode_fused: state %zu < %zu bytes", state_bytes,
               sd_decode_state_bytes(B, M, N, h, w));
    const size_t need = sd_decode_fused_workspace_bytes(B, M, N, h, w, K, P);
    SD_REQUIRE(workspace_bytes >= need, SD_ERR_WORKSPACE, "sd_decode_fused: workspace %zu < %zu", workspace_bytes, need);
    // rank sort needs 2 * np2 <= cap with np2 >= max(K, P): never below 2 * next_pow2(max(K, P))
    int sort_cap = exact_topk ? FUSED_CAP_EXACT : FUSED_CAP_FAST;
    while (sort_cap < 2 * std::max(K, P)) sort_cap *= 2;
    const FusedLds lds = fused_lds(K, P, (int)nti, sort_cap, th);
    SD_REQUIRE(lds.total <= FUSED_LDS_LIMIT, SD_ERR_INVALID, "sd_decode_fused: %d maps x %d tiles need %d bytes of LDS (> %d); use sd_decode",
               M + N, tiles, lds.total, FUSED_LDS_LIMIT);
    auto kern = th == 32 ? k_decode_fused<32> : k_decode_fused<16>;
    if (lds.total > 48 * 1024) {
        static thread_local bool raised[2] = {false, false};      // per host thread: cheap, idempotent
        if (!raised[th == 32]) {
            SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS_LIMIT));
            raised[th == 32] = true;
        }
    }
    Group g0{anchor_hm, a_sb, a_sc, M}, g1{part_hm, p_sb, p_sc, N};
    RegMaps rm{offsets, o_sb, o_sc, embeddings, e_sb, e_sc};
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * nti + B)), dim3(FUSED_THREADS), (size_t)lds.total, (hipStream_t)stream, g0, g1,
                       h, w, tiles_x, tiles, exact_topk ? 0.f : conf, reinterpret_cast<uint64_t*>(workspace),
                       reinterpret_cast<unsigned*>(state), K, P, sort_cap, conf, dist_px, rm, packed, B);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_decode_group(const float* a_score, const int64_t* a_ind, const float* a_cls, const float* p_score, const int64_t* p_ind,
                    const float* p_cls, const float* offsets, int64_t o_sb, int64_t o_sc, const float* embeddings, int64_t e_sb,
                    int64_t e_sc, int B, int h, int w, int K, int P, float conf, float dist_px, void* packed, sd_stream_t stream) {
    if (int e = check_map("sd_decode_group(offsets)", offsets, o_sb, o_sc, B, 2, h, w)) return e;
    if (int e = check_map("sd_decode_group(embeddings)", embeddings, e_sb, e_sc, B, 2, h, w)) return e;
    SD_REQUIRE(a_score && a_ind && a_cls && p_score && p_ind && p_cls && packed, SD_ERR_INVALID, "sd_decode_group: null pointer");
    SD_REQUIRE(K > 0 && K <= SD_MAX_TOPK && P > 0 && P <= SD_MAX_TOPK, SD_ERR_INVALID, "sd_decode_group: K/P out of range");
    RegMaps rm{offsets, o_sb, o_sc, embeddings, e_sb, e_sc};
    hipLaunchKernelGGL(k_group_only, dim3(B), dim3(SEL_THREADS), 0, (hipStream_t)stream, a_score, a_ind, a_cls, p_score, p_ind, p_cls,
                       h * w, w, K, P, conf, dist_px, rm, packed, B);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
