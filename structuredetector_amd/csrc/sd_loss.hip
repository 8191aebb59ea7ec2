// SDNet loss forward / backward on gfx950.
// Follows Loss.forward, src/sdnet/model/loss.py:17-50; L1Loss :53-64; FocalLoss :91-117;
// nn.MSELoss (mean) :13; clamped_sigmoid src/sdnet/utils/utils.py:355-361.
// Forward = one fused map-reduce pass over the (pred, target) heatmaps (HBM-bound: both maps are
// read once) + a single-block finalize that also evaluates the three masked-L1 terms.  The
// reference's host branches (`numel == 0`, `num_pos == 0`) are evaluated on the device, so the
// training step has no host sync.  Reductions are deterministic (fixed tree, no float atomics).
#include "sd_common.h"

namespace sd {

constexpr int LOSS_CHUNK = 4096;   // elements of one (b, c) plane handled by one 256-thread block

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

struct HmView {
    const float* x;  int64_t x_sb, x_sc;
    const float* t;  int64_t t_sb, t_sc;
};

// partial[(b*C + c)*chunks + chunk] = {sum0, sum1, sum2}
//   mse  : sum0 = sum (s - t)^2
//   focal: sum0 = sum log(1-s) s^2 (1-t)^4 [t<1], sum1 = sum log(s) (1-s)^2 [t==1], sum2 = #[t==1]
__global__ __launch_bounds__(256) void k_loss_hm_partial(HmView a, HmView p, int M, int hw, int chunks, int focal,
                                                          float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.z, c = blockIdx.y, chunk = blockIdx.x;
    const int C = gridDim.y;
    const HmView v = (c < M) ? a : p;
    const int cc = (c < M) ? c : c - M;
    const float* x = v.x + (int64_t)b * v.x_sb + (int64_t)cc * v.x_sc;
    const float* t = v.t + (int64_t)b * v.t_sb + (int64_t)cc * v.t_sc;
    const int beg = chunk * LOSS_CHUNK, end = min(beg + LOSS_CHUNK, hw);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
        const float4 xv = *reinterpret_cast<const float4*>(x + i);
        const float4 tv = *reinterpret_cast<const float4*>(t + i);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float ts[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = clamped_sigmoid(xs[j]);
            const float tt = ts[j];
            if (!focal) {
                const float d = s - tt;
                s0 += d * d;
            } else {
                const float om = 1.0f - s;
                if (tt < 1.0f) {
                    const float w1 = 1.0f - tt, w2 = w1 * w1;
                    s0 += logf(om) * (s * s) * (w2 * w2);
                }
                if (tt == 1.0f) {
                    s1 += logf(s) * (om * om);
                    s2 += 1.0f;
                }
            }
        }
    }
    s0 = block_sum_256(s0, red);
    s1 = block_sum_256(s1, red);
    s2 = block_sum_256(s2, red);
    if (threadIdx.x == 0) {
        float* dst = partial + (((int64_t)b * C + c) * chunks + chunk) * 3;
        dst[0] = s0; dst[1] = s1; dst[2] = s2;
    }
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

struct RegView {
    const float* f;  int64_t sb, sc;     // (B,2,h*w)
    const float* tgt;                    // (B,n,2)
    const int64_t* inds;                 // (B,n)
    const uint8_t* mask;                 // (B,n)
    int n;
};

// sum |pred - target| * mask  and  #mask   (loss.py:58-64)
__device__ void l1_term(const RegView& r, int B, int64_t hw, double* red, double& sum, double& cnt) {
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < B * r.n; i += 256) {
        if (!r.mask[i]) continue;
        const int b = i / r.n;
        int64_t id = r.inds[i];
        id = id < 0 ? 0 : (id >= hw ? hw - 1 : id);
        const float* f = r.f + (int64_t)b * r.sb + id;
        s += (double)fabsf(f[0] - r.tgt[2 * i]) + (double)fabsf(f[r.sc] - r.tgt[2 * i + 1]);
        c += 1.0;
    }
    sum = block_sum_d(s, red);
    cnt = block_sum_d(c, red);
}

__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ partial, int B, int M, int N, int hw, int chunks,
                                                        int focal, float hm_w, float off_w, float emb_w, RegView ra, RegView rp,
                                                        RegView re, float* __restrict__ out8) {
    __shared__ double red[4];
    const int C = M + N;
    double acc[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (int i = threadIdx.x; i < B * C * chunks; i += 256) {
        const int c = (i / chunks) % C;
        const int g = (c < M) ? 0 : 1;
        acc[g][0] += partial[3 * (int64_t)i + 0];
        acc[g][1] += partial[3 * (int64_t)i + 1];
        acc[g][2] += partial[3 * (int64_t)i + 2];
    }
    double tot[2][3];
    for (int g = 0; g < 2; ++g)
        for (int j = 0; j < 3; ++j) tot[g][j] = block_sum_d(acc[g][j], red);
    double la, ca, lp, cp, le, ce;
    l1_term(ra, B, hw, red, la, ca);
    l1_term(rp, B, hw, red, lp, cp);
    l1_term(re, B, hw, red, le, ce);
    if (threadIdx.x == 0) {
        double hm = 0.0;
        for (int g = 0; g < 2; ++g) {
            const double n_el = (double)B * (g ? N : M) * hw;
            if (!focal) hm += tot[g][0] / n_el;                                     // nn.MSELoss mean
            else hm += (tot[g][2] == 0.0) ? -tot[g][0] : -(tot[g][1] + tot[g][0]) / tot[g][2];   // loss.py:110-117
        }
        hm *= hm_w;
        const double off = off_w * ((ca > 0 ? la / ca : 0.0) + (cp > 0 ? lp / cp : 0.0));   // loss.py:26-39
        const double emb = emb_w * (ce > 0 ? le / ce : 0.0);                                // loss.py:41-46
        out8[0] = (float)(hm + off + emb);
        out8[1] = (float)hm; out8[2] = (float)off; out8[3] = (float)emb;
        out8[4] = (float)tot[0][2]; out8[5] = (float)tot[1][2];
        out8[6] = (float)ca; out8[7] = (float)cp;
    }
}

// d total / d logits for the heatmap channels; zero for the 4 regression channels (filled in by
// k_loss_bwd_scatter afterwards).  grid (chunks, M+N+4, B).
__global__ __launch_bounds__(256) void k_loss_bwd_dense(HmView a, HmView p, int B, int M, int N, int hw, int focal, float hm_w,
                                                         const float* __restrict__ out8, const float* __restrict__ grad_out,
                                                         float* __restrict__ dhead) {
    const int b = blockIdx.z, c = blockIdx.y, chunk = blockIdx.x;
    const int C = M + N + 4;
    float* dst = dhead + ((int64_t)b * C + c) * hw;
    const int beg = chunk * LOSS_CHUNK, end = min(beg + LOSS_CHUNK, hw);
    if (c >= M + N) {
        for (int i = beg + threadIdx.x * 4; i < end; i += 1024) *reinterpret_cast<float4*>(dst + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int g = (c < M) ? 0 : 1;
    const HmView v = g ? p : a;
    const int cc = g ? c - M : c;
    const float* x = v.x + (int64_t)b * v.x_sb + (int64_t)cc * v.x_sc;
    const float* t = v.t + (int64_t)b * v.t_sb + (int64_t)cc * v.t_sc;
    const float go = grad_out[0] * hm_w;
    const float npos = out8[4 + g];
    float scale;   // multiplies d(sum)/ds
    if (!focal) scale = go * 2.0f / ((float)B * (float)(g ? N : M) * (float)hw);
    else scale = (npos == 0.f) ? -go : -go / npos;
    for (int i = beg + threadIdx.x * 4; i < end; i += 1024) {
        const float4 xv = *reinterpret_cast<const float4*>(x + i);
        const float4 tv = *reinterpret_cast<const float4*>(t + i);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
        const float ts[4] = {tv.x, tv.y, tv.z, tv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sr = sigmoid_raw(xs[j]);
            const bool pass = (sr >= kClampLo) && (sr <= kClampHi);       // clamp backward (inclusive bounds)
            const float s = fminf(fmaxf(sr, kClampLo), kClampHi);
            const float tt = ts[j];
            float dls;                                                     // d(sum term)/ds
            if (!focal) {
                dls = s - tt;
            } else {
                const float om = 1.0f - s;
                dls = 0.f;
                if (tt < 1.0f) {
                    const float w1 = 1.0f - tt, w2 = w1 * w1;
                    dls += (2.0f * s * logf(om) - s * s / om) * (w2 * w2);
                }
                if (tt == 1.0f && npos != 0.f) dls += om * om / s - 2.0f * om * logf(s);
            }
            o[j] = pass ? scale * dls * (sr * (1.0f - sr)) : 0.f;
        }
        *reinterpret_cast<float4*>(dst + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Gather-backward of the three L1 terms = scatter-add of sign(pred - target) * w / #valid into
// the regression channels.  One block per image; duplicates (several keypoints in one cell) are
// summed in list order by the first entry of each duplicate set -> plain stores, deterministic.
__global__ __launch_bounds__(256) void k_loss_bwd_scatter(RegView ra, RegView rp, RegView re, int M, int N, int64_t hw, float off_w,
                                                           float emb_w, const float* __restrict__ out8,
                                                           const float* __restrict__ grad_out, float* __restrict__ dhead) {
    const int b = blockIdx.x;
    const int C = M + N + 4;
    const float go = grad_out[0];
    const float ca = out8[6], cp = out8[7];
    const float wa = ca > 0.f ? go * off_w / ca : 0.f;
    const float wp = cp > 0.f ? go * off_w / cp : 0.f;
    const float we = cp > 0.f ? go * emb_w / cp : 0.f;
    const int K = ra.n, P = rp.n;
    auto entry = [&](int e, int64_t& id, float& gx, float& gy) -> bool {   // offsets list: anchors then parts
        const RegView& r = (e < K) ? ra : rp;
        const int i = b * r.n + ((e < K) ? e : e - K);
        if (!r.mask[i]) return false;
        id = r.inds[i];
        id = id < 0 ? 0 : (id >= hw ? hw - 1 : id);
        const float* f = r.f + (int64_t)b * r.sb + id;
        const float dx = f[0] - r.tgt[2 * i], dy = f[r.sc] - r.tgt[2 * i + 1];
        const float w = (e < K) ? wa : wp;
        gx = (dx > 0.f ? w : (dx < 0.f ? -w : 0.f));
        gy = (dy > 0.f ? w : (dy < 0.f ? -w : 0.f));
        return true;
    };
    float* d_off = dhead + ((int64_t)b * C + M + N) * hw;
    for (int e = threadIdx.x; e < K + P; e += 256) {
        int64_t id, id2; float gx, gy, hx, hy;
        if (!entry(e, id, gx, gy)) continue;
        bool first = true;
        for (int j = 0; j < e && first; ++j) if (entry(j, id2, hx, hy) && id2 == id) first = false;
        if (!first) continue;
        for (int j = e + 1; j < K + P; ++j) if (entry(j, id2, hx, hy) && id2 == id) { gx += hx; gy += hy; }
        d_off[id] = gx; d_off[hw + id] = gy;
    }
    auto entry_e = [&](int e, int64_t& id, float& gx, float& gy) -> bool {
        const int i = b * re.n + e;
        if (!re.mask[i]) return false;
        id = re.inds[i];
        id = id < 0 ? 0 : (id >= hw ? hw - 1 : id);
        const float* f = re.f + (int64_t)b * re.sb + id;
        const float dx = f[0] - re.tgt[2 * i], dy = f[re.sc] - re.tgt[2 * i + 1];
        gx = (dx > 0.f ? we : (dx < 0.f ? -we : 0.f));
        gy = (dy > 0.f ? we : (dy < 0.f ? -we : 0.f));
        return true;
    };
    float* d_emb = dhead + ((int64_t)b * C + M + N + 2) * hw;
    for (int e = threadIdx.x; e < re.n; e += 256) {
        int64_t id, id2; float gx, gy, hx, hy;
        if (!entry_e(e, id, gx, gy)) continue;
        bool first = true;
        for (int j = 0; j < e && first; ++j) if (entry_e(j, id2, hx, hy) && id2 == id) first = false;
        if (!first) continue;
        for (int j = e + 1; j < re.n; ++j) if (entry_e(j, id2, hx, hy) && id2 == id) { gx += hx; gy += hy; }
        d_emb[id] = gx; d_emb[hw + id] = gy;
    }
}

static int check_desc(const sd_loss_desc* d) {
    SD_REQUIRE(d != nullptr, SD_ERR_INVALID, "loss: null descriptor");
    SD_REQUIRE(d->B > 0 && d->M > 0 && d->N > 0 && d->h > 0 && d->w > 0 && d->K > 0 && d->P > 0, SD_ERR_INVALID, "loss: bad sizes");
    SD_REQUIRE(((int64_t)d->h * d->w) % 4 == 0, SD_ERR_INVALID, "loss: h*w must be a multiple of 4");
    SD_REQUIRE(d->anchor_hm && d->part_hm && d->offsets && d->embeddings && d->t_anchor_hm && d->t_part_hm && d->anchor_inds &&
                   d->part_inds && d->anchor_offsets && d->part_offsets && d->t_embeddings && d->anchor_mask && d->part_mask,
               SD_ERR_INVALID, "loss: null pointer in descriptor");
    SD_REQUIRE(d->hm_loss_fn == SD_HM_MSE || d->hm_loss_fn == SD_HM_FOCAL, SD_ERR_INVALID, "loss: hm_loss_fn must be mse or focal");
    const void* maps[4] = {d->anchor_hm, d->part_hm, d->t_anchor_hm, d->t_part_hm};
    const int64_t strides[8] = {d->a_sb, d->a_sc, d->p_sb, d->p_sc, d->ta_sb, d->ta_sc, d->tp_sb, d->tp_sc};
    for (int i = 0; i < 4; ++i) SD_REQUIRE(aligned16(maps[i]), SD_ERR_ALIGN, "loss: heatmap pointer %d not 16-byte aligned", i);
    for (int i = 0; i < 8; ++i) SD_REQUIRE(strides[i] % 4 == 0, SD_ERR_ALIGN, "loss: heatmap stride %d not a multiple of 4", i);
    return 0;
}

}  // namespace sd

using namespace sd;

extern "C" {

size_t sd_loss_workspace_bytes(int B, int M, int N, int h, int w) {
    const int chunks = cdiv((int64_t)h * w, LOSS_CHUNK);
    return align_up((size_t)B * (M + N) * chunks * 3 * sizeof(float), 256);
}

int sd_loss_fwd(const sd_loss_desc* d, float* out8, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_desc(d)) return e;
    SD_REQUIRE(out8 && workspace, SD_ERR_INVALID, "sd_loss_fwd: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_loss_workspace_bytes(d->B, d->M, d->N, d->h, d->w), SD_ERR_WORKSPACE, "sd_loss_fwd: workspace too small");
    const int hw = d->h * d->w, chunks = cdiv(hw, LOSS_CHUNK);
    const int focal = d->hm_loss_fn == SD_HM_FOCAL;
    HmView a{d->anchor_hm, d->a_sb, d->a_sc, d->t_anchor_hm, d->ta_sb, d->ta_sc};
    HmView p{d->part_hm, d->p_sb, d->p_sc, d->t_part_hm, d->tp_sb, d->tp_sc};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_hm_partial, dim3(chunks, d->M + d->N, d->B), dim3(256), 0, st, a, p, d->M, hw, chunks, focal,
                       (float*)workspace);
    SD_LAUNCH_CHECK();
    RegView ra{d->offsets, d->o_sb, d->o_sc, d->anchor_offsets, d->anchor_inds, d->anchor_mask, d->K};
    RegView rp{d->offsets, d->o_sb, d->o_sc, d->part_offsets, d->part_inds, d->part_mask, d->P};
    RegView re{d->embeddings, d->e_sb, d->e_sc, d->t_embeddings, d->part_inds, d->part_mask, d->P};
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, st, (const float*)workspace, d->B, d->M, d->N, hw, chunks, focal,
                       d->hm_weight, d->offset_weight, d->embedding_weight, ra, rp, re, out8);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_loss_bwd(const sd_loss_desc* d, const float* out8, const float* grad_out, float* dhead, sd_stream_t stream) {
    if (int e = check_desc(d)) return e;
    SD_REQUIRE(out8 && grad_out && dhead, SD_ERR_INVALID, "sd_loss_bwd: null pointer");
    SD_REQUIRE(aligned16(dhead), SD_ERR_ALIGN, "sd_loss_bwd: dhead must be 16-byte aligned");
    const int hw = d->h * d->w, chunks = cdiv(hw, LOSS_CHUNK);
    const int focal = d->hm_loss_fn == SD_HM_FOCAL;
    HmView a{d->anchor_hm, d->a_sb, d->a_sc, d->t_anchor_hm, d->ta_sb, d->ta_sc};
    HmView p{d->part_hm, d->p_sb, d->p_sc, d->t_part_hm, d->tp_sb, d->tp_sc};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_bwd_dense, dim3(chunks, d->M + d->N + 4, d->B), dim3(256), 0, st, a, p, d->B, d->M, d->N, hw, focal,
                       d->hm_weight, out8, grad_out, dhead);
    SD_LAUNCH_CHECK();
    RegView ra{d->offsets, d->o_sb, d->o_sc, d->anchor_offsets, d->anchor_inds, d->anchor_mask, d->K};
    RegView rp{d->offsets, d->o_sb, d->o_sc, d->part_offsets, d->part_inds, d->part_mask, d->P};
    RegView re{d->embeddings, d->e_sb, d->e_sc, d->t_embeddings, d->part_inds, d->part_mask, d->P};
    hipLaunchKernelGGL(k_loss_bwd_scatter, dim3(d->B), dim3(256), 0, st, ra, rp, re, d->M, d->N, (int64_t)hw, d->offset_weight,
                       d->embedding_weight, out8, grad_out, dhead);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
