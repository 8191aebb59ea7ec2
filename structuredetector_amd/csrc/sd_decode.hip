// Decoder hot path for gfx950: clamped sigmoid + 5x5 NMS + peak compaction, exact top-k
// selection, offset/embedding gather and anchor<->part association.
// Follows src/sdnet/data/decoders.py:41-100 and src/sdnet/utils/utils.py:341-361,422-467 of
// the reference.  All arithmetic that decides an index is done exactly as the reference's fp32
// tensor ops: separately rounded mul/add/sqrt -- floating-point contraction is OFF in this
// file.
#pragma clang fp contract(off)
#include "sd_common.h"

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

// ---------------------------------------------------------------------------------------------
// Kernel 1: tile NMS.  One 256-thread block per 64x16 output tile of one map; the (64+4)x(16+4)
// neighbourhood is staged once in LDS as clamped-sigmoid values, 5-max is separable
// (row pass into a second LDS array, column pass in registers).  HBM-bound: every logit is read
// once (+ halo re-reads, 1.33x, served by L2).
//   MODE 0: dense output  out = keep ? v : 0            (nms(), utils.py:441-443)
//   MODE 1: compaction    survivors appended as keys to the per-(image, group) candidate list
// ---------------------------------------------------------------------------------------------
constexpr int TW = 64, TH = 16, HALO = 2;
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
constexpr int SEL_THREADS = 512;      // threads of one selection "team" (8 waves)
constexpr int SORT_CAP = 4096;

// One team = SEL_THREADS threads working on one candidate list.  k_select_group runs two teams in one
// 1024-thread block (anchors and parts side by side): every team executes the SAME sequence of block barriers
// (the sequence depends only on `np2` / `use_radix`, which the caller makes identical for both teams).
struct Team {
    int tid;            // 0 .. SEL_THREADS-1 inside the team
    uint64_t* buf;      // [SORT_CAP]
    int* hist;          // [2][256]  (double-buffered by radix pass)
    int* misc;          // [4]
    int* flags;         // [SD_MAX_TOPK]
    int team;           // index of this team inside the block
    int* alive;         // [2], shared by ALL teams of the block: 1 while a team still needs radix passes
};

// In-place descending bitonic sort of buf[0..np2).  Each wave owns a contiguous range of R elements; stages
// whose compare distance stays inside a range need no block barrier (LDS operations of one wave execute in
// order), only the few long-distance stages synchronise the whole block: 6 instead of 66 barriers for 2048 keys.
__device__ void bitonic_desc(const Team& T, int np2) {
    const int lane = T.tid & 63, wave = T.tid >> 6;
    const int R = max(np2 >> 3, 128);                 // elements per wave range
    const int half_pairs = min(R, np2) >> 1;          // compare-exchange pairs per range and stage
    const bool active = wave * R < np2;
    bool local_dirty = false;                         // wave-local stages since the last block barrier
    for (int k = 2; k <= np2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (2 * j <= R) {                         // partner inside the wave's own range
                if (active) {
                    const int base = wave * R;
                    for (int t = lane; t < half_pairs; t += 64) {
                        const int i = base + 2 * t - (t & (j - 1)), l = i + j;
                        const uint64_t a = T.buf[i], bb = T.buf[l];
                        if ((a < bb) == ((i & k) == 0)) { T.buf[i] = bb; T.buf[l] = a; }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // wave-level ordering of the LDS traffic
                local_dirty = true;
            } else {
                if (local_dirty) { __syncthreads(); local_dirty = false; }
                for (int t = T.tid; t < (np2 >> 1); t += SEL_THREADS) {
                    const int i = 2 * t - (t & (j - 1)), l = i + j;
                    const uint64_t a = T.buf[i], bb = T.buf[l];
                    if ((a < bb) == ((i & k) == 0)) { T.buf[i] = bb; T.buf[l] = a; }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Exact top-k of n unique keys (descending) into T.buf[0..k).  use_radix / np2 are team-uniform AND identical for
// all teams of the block.  !use_radix: n <= SORT_CAP, everything is sorted in LDS (np2 >= max(n, k)).
// use_radix: MSB-first 8-bit radix select over the global list finds the k-th largest key, the k keys >= it are
// collected and sorted (np2 >= k).  Slots beyond min(n, k) are left as key 0 (filled by the caller).
__device__ void team_select_topk(const Team& T, const uint64_t* __restrict__ cand, int n, int k, bool use_radix, int np2) {
    const int tid = T.tid;
    if (!use_radix) {
        for (int i = tid; i < np2; i += SEL_THREADS) T.buf[i] = (i < n) ? cand[i] : 0ull;
        __syncthreads();
        bitonic_desc(T, np2);
        return;
    }
    // MSB-first 8-bit radix select.  Two block barriers per pass (histograms double-buffered by pass parity, bucket scan by
    // one wave), key loads batched four deep, and the passes stop as soon as the boundary bucket is taken whole in EVERY team
    // of the block (unique keys: usually after the score bytes) -- `alive` is block-wide so that all teams leave together.
    uint64_t prefix = 0, mask = 0;
    int remaining = min(k, n);
    bool done = remaining == 0;
    if (done) prefix = ~0ull;
    static_assert(SEL_THREADS == 512, "one thread per entry of the double-buffered histogram");
    T.hist[tid] = 0;
    if (tid == 0) T.alive[T.team] = done ? 0 : 1;
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
        int* hcur = T.hist + (pass & 1) * 256;
        int* hnext = T.hist + ((pass & 1) ^ 1) * 256;
        const int shift = pass * 8;
        if (!done) {
            for (int base = tid; base < n; base += 4 * SEL_THREADS) {
                uint64_t key[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) key[u] = (base + u * SEL_THREADS < n) ? cand[base + u * SEL_THREADS] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (base + u * SEL_THREADS < n && (key[u] & mask) == prefix) atomicAdd(&hcur[(int)((key[u] >> shift) & 255ull)], 1);
            }
        }
        if (tid < 256) hnext[tid] = 0;
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
        if ((T.alive[0] | T.alive[1]) == 0) break;
    }
    // keys are unique, so exactly min(k, n) keys are >= prefix
    for (int i = tid; i < np2; i += SEL_THREADS) T.buf[i] = 0ull;
    if (tid == 0) T.misc[2] = 0;
    __syncthreads();
    for (int base = tid; base < n; base += 4 * SEL_THREADS) {
        uint64_t key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) key[u] = (base + u * SEL_THREADS < n) ? cand[base + u * SEL_THREADS] : 0ull;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (base + u * SEL_THREADS < n && key[u] >= prefix) {
                const int slot = atomicAdd(&T.misc[2], 1);
                if (slot < np2) T.buf[slot] = key[u];
            }
    }
    __syncthreads();
    bitonic_desc(T, np2);
}

// Suppressed pixels have score exactly 0; when fewer than k peaks exist the reference's remaining top-k slots
// are zeros (utils.py:451 on the NMS'ed map).  Fill them with the lowest class-major flat indices that are not
// peaks (stable order).  Always executes two block barriers (team-uniform control flow).
__device__ void fill_zero_slots(const Team& T, int npos, int k) {
    const int tid = T.tid;
    if (npos < k) {
        for (int f = tid; f < k; f += SEL_THREADS) {
            int used = 0;
            for (int j = 0; j < npos; ++j) used |= ((uint32_t)(~T.buf[j]) == (uint32_t)f);
            T.flags[f] = used ? 0 : 1;
        }
    }
    __syncthreads();
    if (npos < k) {
        for (int f = tid; f < k; f += SEL_THREADS) {
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
    const Team T{(int)threadIdx.x, buf, hist, misc, flags, 0, alive};
    const bool use_radix = n > SORT_CAP;
    const int np2 = max(next_pow2(use_radix ? k : max(n, k)), 2);
    team_select_topk(T, cand + (int64_t)b * cap, n, k, use_radix, np2);
    if (do_fill) fill_zero_slots(T, min(n, k), k);
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
    L.assign = reinterpret_cast<int*>(f);
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
    for (int a = tid; a < K; a += (int)blockDim.x) {
        const int ind = ai_[a];
        const int y = ind / w, x = ind - y * w;
        const float score = as_[a];
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
    __syncthreads();
    for (int p = tid; p < P; p += (int)blockDim.x) {
        const int ind = pi_[p];
        const int y = ind / w, x = ind - y * w;
        const float score = ps_[p];
        const float ex = emb_b[ind], ey = emb_b[rm.e_sc + ind]; // decoders.py:66
        const float px = (float)x + off_b[ind];                 // decoders.py:67
        const float py = (float)y + off_b[rm.o_sc + ind];       // decoders.py:68
        const float ox = px + ex, oy = py + ey;                 // decoders.py:69-70
        const bool m = score > conf;                            // decoders.py:78
        const float orx = m ? ox : -1e6f, ory = m ? oy : -1e6f; // decoders.py:80-81
        float best = INFINITY;
        int best_a = 0;
        for (int a = 0; a < K; ++a) {                           // decoders.py:88-98, utils.py:433-435
            const float dx = orx - posx[a], dy = ory - posy[a];
            const float sx = dx * dx, sy = dy * dy;
            const float d = sqrtf(sx + sy);
            if (d < best) { best = d; best_a = a; }             // strict <: lowest anchor rank wins ties
        }
        float* po = L.part_out + ((int64_t)b * P + p) * 6;
        po[0] = px; po[1] = py; po[2] = score; po[3] = (float)pc_[p]; po[4] = ox; po[5] = oy;
        L.part_emb[((int64_t)b * P + p) * 2 + 0] = ex;
        L.part_emb[((int64_t)b * P + p) * 2 + 1] = ey;
        L.part_smask[(int64_t)b * P + p] = m ? score : -1.0f;   // decoders.py:79
        L.part_ind[(int64_t)b * P + p] = ind;
        L.assign[(int64_t)b * P + p] = (best < dist_px) ? best_a : -1;   // decoders.py:100
    }
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
    const Team T{tid, buf[team], hist[team], misc[team], flags[team], team, alive};

    const int n0 = counters[(b * 2 + 0) * CNT_STRIDE], n1 = counters[(b * 2 + 1) * CNT_STRIDE];
    // identical barrier sequence for both teams: the path and the sort size come from the larger list
    const bool use_radix = max(n0, n1) > SORT_CAP;
    const int np2 = max(next_pow2(use_radix ? max(K, P) : max(max(n0, n1), max(K, P))), 2);
    const int n = team ? n1 : n0, k = team ? P : K;
    const uint64_t* cand = team ? cand1 + (int64_t)b * N * hw : cand0 + (int64_t)b * M * hw;
    team_select_topk(T, cand, n, k, use_radix, np2);
    fill_zero_slots(T, min(n, k), k);
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
    char* p = reinterpret_cast<char*>(ws);
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

size_t sd_decode_workspace_bytes(int B, int M, int N, int h, int w, int K, int P) {
    (void)K; (void)P;
    return carve(nullptr, B, M, N, h, w).bytes;
}

size_t sd_decode_packed_words(int B, int K, int P) { return (size_t)B * (6 * (size_t)K + 11 * (size_t)P); }

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
    const PeaksWs ws = carve(workspace, B, M, N, h, w);
    SD_REQUIRE(workspace_bytes >= ws.bytes, SD_ERR_WORKSPACE, "sd_decode: workspace %zu < %zu", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    SD_HIP(hipMemsetAsync(ws.counters, 0, (size_t)B * 2 * CNT_STRIDE * sizeof(int), st));
    const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
    Group g0{anchor_hm, a_sb, a_sc, M}, g1{part_hm, p_sb, p_sc, N};
    hipLaunchKernelGGL(k_nms_tile<1>, dim3(tiles_x * tiles_y, M + N, B), dim3(256), 0, st, g0, g1, h, w, tiles_x, 1, exact_topk ? 0.f : conf, (float*)nullptr,
                       ws.cand0, ws.cand1, ws.counters);
    SD_LAUNCH_CHECK();
    RegMaps rm{offsets, o_sb, o_sc, embeddings, e_sb, e_sc};
    hipLaunchKernelGGL(k_select_group, dim3(B), dim3(2 * SEL_THREADS), 0, st, ws.cand0, ws.cand1, ws.counters, M, N, h, w, K, P, conf,
                       dist_px, rm, packed, B);
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
