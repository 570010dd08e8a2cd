// Standard blank-CTC loss + gradient for gfx950 (BASELINE config 5: long sequences).
//
// Semantics: torch.nn.CTCLoss(blank, reduction='mean', zero_infinity=False) as the
// reference uses it at models/layers/AsyncTFCriterion.py:198,319-321 (arithmetic lives in
// aten::_ctc_loss, not in the reference repo): extended label l' of n = 2L+1 states,
//   alpha_t(s) = LSE(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != blank
//                and l'_s != l'_{s-2}]) + lp_t(l'_s),      nll = -LSE(alpha_{T-1}(n-1), alpha_{T-1}(n-2)),
//   grad[t,c] = (exp(lp[t,c]) - sum_{s: l'_s = c} gamma_t(s)) * grad_scale / max(L,1), 0 for t >= T_b.
//
// T x (2S+1) does not fit in LDS (config 5: 2000 x 201), so the lattice lives in the
// caller's workspace and the work is split where the parallelism changes:
//   K0 gather   (wide, all CUs)  em[b,t,s] = lp[t,b,l'_s]: rows of lp are read whole and
//                                coalesced once; per-sample state tables are built here
//   K1 chains   (one WG / sample) wave 0 = alpha, wave 1 = beta, K states per lane, the two
//                                neighbour states through DPP wave shifts, emission rows
//                                prefetched 8 deep from the workspace; writes alpha, beta
//   K2 grad     (wide)           one wave per (t,b) row: gamma_t = softmax_s(alpha+beta-e)
//                                (row-normalised, see lattice.hpp), folded per class
//                                (blank by a wave reduction, repeated labels by the
//                                first-occurrence chain), then the dense row
//                                exp(lp) - occupancy with coalesced loads / stores.
#include "common.hpp"
#include "launch.hpp"

namespace ctc {

constexpr float kNegB = -1.0e30f;    // finite stand-in for -inf inside the scans
constexpr int kDepth = 8;            // emission rows in flight ahead of a chain

struct BlankParams {
    const float *lp;
    int64_t st, sb;
    const void *tgt;
    int tgt64;
    const int64_t *in_len, *tgt_len;
    int T, B, C, S, NSP, blank;      // NSP = 64*K padded state count
    float loss_scale, grad_scale;
    float *nll, *loss, *grad;
    unsigned *counter;
    float *em, *al, *be;             // [B][T][NSP] each
    int *cls, *nxt, *first;          // [B][NSP]: class of state s; next state with the same class;
                                     // 1 when s is the first state carrying its (non-blank) class
};

__device__ __forceinline__ bool blank_sample_ok(const BlankParams &p, int b, int &Tb, int &L)
{
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const bool ok = L64 >= 0 && L64 <= p.S && Tb64 >= 0 && Tb64 <= p.T;
    Tb = ok ? (int)Tb64 : 0;
    L = ok ? (int)L64 : 0;
    return ok;
}

// ---- K0: state tables + emission gather -------------------------------------------------
__global__ __launch_bounds__(256) void blank_gather_kernel(BlankParams p, int rows_per_block)
{
    extern __shared__ int s_cls[];                           // [NSP]
    const int b = blockIdx.y, tid = threadIdx.x;
    int Tb, L;
    blank_sample_ok(p, b, Tb, L);
    const int n = 2 * L + 1;
    for (int s = tid; s < p.NSP; s += blockDim.x) {
        int c = p.blank;
        if (s < n && (s & 1)) {
            c = load_label(p.tgt, p.tgt64, (int64_t)b * p.S + (s >> 1));
            c = c < 0 ? 0 : (c >= p.C ? p.C - 1 : c);        // memory safety for bad labels
        }
        s_cls[s] = c;
    }
    __syncthreads();
    if (blockIdx.x == 0)
        for (int s = tid; s < p.NSP; s += blockDim.x) {
            p.cls[b * p.NSP + s] = s_cls[s];
            int nx = -1;
            if (s < n && (s & 1))
                for (int s2 = s + 2; s2 < n; s2 += 2)
                    if (s_cls[s2] == s_cls[s]) { nx = s2; break; }
            p.nxt[b * p.NSP + s] = nx;
            int fi = (s < n && (s & 1)) ? 1 : 0;
            for (int s2 = 1; fi && s2 < s; s2 += 2)
                if (s_cls[s2] == s_cls[s]) fi = 0;
            p.first[b * p.NSP + s] = fi;
        }
    const int t_begin = blockIdx.x * rows_per_block;
    const int t_end = min(t_begin + rows_per_block, Tb);
    for (int t = t_begin; t < t_end; ++t) {
        const float *row = p.lp + (int64_t)t * p.st + (int64_t)b * p.sb;
        float *out = p.em + ((int64_t)b * p.T + t) * p.NSP;
        for (int s = tid; s < p.NSP; s += blockDim.x) out[s] = s < n ? fmaxf(row[s_cls[s]] * kLog2e, kNegB) : kNegB;
    }
}

// ---- K1: alpha / beta chains ----------------------------------------------------------------
// The blank lattice (em, alpha, beta) is kept in LOG2 units: v_exp_f32 / v_log_f32 are base-2,
// so a state update is max, subtract, exp2, add, log2, add with no base-conversion multiplies.
// Only differences alpha+beta-em and the final likelihood (x ln 2) leave the lattice.
__device__ __forceinline__ float lse3_2(float a, float b, float c)
{
    const float m = fmaxf(fmaxf(a, b), c);
    const float s = __builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) + __builtin_amdgcn_exp2f(c - m);
    return m + __builtin_amdgcn_logf(s);
}
__device__ __forceinline__ float lse2_2(float a, float b)
{
    const float m = fmaxf(a, b);
    return m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}

template <int K, bool FWD>
__device__ __forceinline__ void blank_chain(const BlankParams &p, int b, int Tb, int L, float (&a)[K])
{
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1;
    const float *em = p.em + (int64_t)b * p.T * p.NSP + s0;
    float *out = (FWD ? p.al : p.be) + (int64_t)b * p.T * p.NSP + s0;
    const int *cls = p.cls + b * p.NSP;
    bool skip[K], valid[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        valid[k] = s < n;
        // alpha: from s-2 when l'_s is a label differing from l'_{s-2};  beta: from s+2 likewise
        const int s2 = FWD ? s - 2 : s + 2;
        skip[k] = (s & 1) && s2 >= 0 && s2 < n && cls[s] != cls[s2];
    }
    auto row_of = [&](int i) { return FWD ? i : Tb - 1 - i; };
    auto fetch = [&](float (&dst)[K], int i) {
        const float *r = em + (int64_t)row_of(i < Tb ? i : Tb - 1) * p.NSP;
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = r[k];
    };
    auto store = [&](int i) {
        float *r = out + (int64_t)row_of(i) * p.NSP;
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = a[k];
    };
    auto step = [&](int i, const float (&e)[K]) {
        // neighbour lane's two edge states (alpha: previous lane's last two, beta: next lane's first two)
        const float n1 = FWD ? wave_shr1(a[K - 1], kNegB) : wave_shl1(a[0], kNegB);
        const float n2 = FWD ? wave_shr1(a[K - 2], kNegB) : wave_shl1(a[1], kNegB);
        float nx[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float x1, x2;
            if (FWD) {
                x1 = k >= 1 ? a[k - 1] : n1;
                x2 = k >= 2 ? a[k - 2] : (k == 1 ? n1 : n2);
            } else {
                x1 = k + 1 < K ? a[k + 1] : n1;
                x2 = k + 2 < K ? a[k + 2] : (k + 1 < K ? n1 : n2);
            }
            // s = lane*K + k with K even: k even <=> blank state (two predecessors, no skip).
            // States beyond n carry the sentinel emission and just sink (stay finite: they lose
            // 1e30 per step, fp32 holds that for any T); no per-state masking or clamping.
            nx[k] = ((k & 1) ? lse3_2(a[k], x1, skip[k] ? x2 : kNegB) : lse2_2(a[k], x1)) + e[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = nx[k];
        store(i);
    };

    float ring[kDepth][K];
    {   // first row: the two entry states only
        float e0[K];
        fetch(e0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int s = s0 + k;
            const bool entry = FWD ? (s == 0 || s == 1) : (s == n - 1 || s == n - 2);
            a[k] = (entry && valid[k]) ? e0[k] : kNegB;
        }
        store(0);
    }
#pragma unroll
    for (int j = 0; j < kDepth; ++j) fetch(ring[j], 1 + j);
    int i = 1;
    for (; i + kDepth <= Tb; i += kDepth) {
#pragma unroll
        for (int j = 0; j < kDepth; ++j) {
            float e[K];
#pragma unroll
            for (int k = 0; k < K; ++k) e[k] = ring[j][k];
            fetch(ring[j], i + j + kDepth);
            step(i + j, e);
        }
    }
#pragma unroll
    for (int j = 0; j < kDepth; ++j)
        if (i + j < Tb) step(i + j, ring[j]);
}

template <int K>
__global__ __launch_bounds__(128) void blank_chain_kernel(BlankParams p)
{
    const int b = blockIdx.x, w = wave_id();
    int Tb, L;
    const bool ok = blank_sample_ok(p, b, Tb, L);
    const int n = 2 * L + 1;
    float a[K];
    if (w == 0) {
        float nll = __builtin_inff();
        if (ok && Tb > 0) {
            blank_chain<K, true>(p, b, Tb, L, a);
            float v1 = 0.f, v2 = 0.f;                       // alpha_{T-1}(n-1), alpha_{T-1}(n-2)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int s = lane_id() * K + k;
                if (s == n - 1) v1 = a[k];
                if (s == n - 2) v2 = a[k];
            }
            v1 = wave_sum(v1);
            v2 = n >= 2 ? wave_sum(v2) : kNegB;
            const float ll2 = lse2_2(v1, v2);
            nll = ll2 < -1.0e29f ? __builtin_inff() : -ll2 * kLn2;
        } else if (ok && L == 0) {
            nll = 0.f;                                       // empty input, empty target
        }
        publish_and_reduce(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter,
                           [&](float v, int i) {
                               const int64_t Li = p.tgt_len[i];
                               return v / (float)(Li > 1 ? Li : 1);
                           });
    } else if (w == 1 && p.grad && ok && Tb > 0) {
        blank_chain<K, false>(p, b, Tb, L, a);
    }
}

// ---- K2: gamma -> gradient rows ----------------------------------------------------------------
constexpr int kGradWaves = 4;

constexpr int kMaxV4 = 4;

// One (t,b) row of work for a wave of blank_grad_kernel: everything it loads from HBM.
template <int K>
struct BlankRow {
    float4 xr[kMaxV4];
    float al[K], be[K], em[K];
    int t, b, Tb, L;
    bool live;
};

template <int K, bool VEC4>
__device__ __forceinline__ void blank_row_load(const BlankParams &p, int idx, BlankRow<K> &r)
{
    const int lane = lane_id();
    r.t = idx / p.B;
    r.b = idx - r.t * p.B;                                   // consecutive waves -> consecutive b: contiguous rows
    const bool ok = blank_sample_ok(p, r.b, r.Tb, r.L);
    r.live = ok && r.t < r.Tb && p.nll[r.b] < 3.0e38f;       // beyond T_b, or no alignment: zero row
    if (!r.live) return;                                     // wave-uniform
    const int64_t off = ((int64_t)r.b * p.T + r.t) * p.NSP + lane * K;
#pragma unroll
    for (int k = 0; k < K; ++k) { r.al[k] = p.al[off + k]; r.be[k] = p.be[off + k]; r.em[k] = p.em[off + k]; }
    if (VEC4) {
        const float *row = p.lp + (int64_t)r.t * p.st + (int64_t)r.b * p.sb;
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) {
            const int q = lane + kWave * i;
            r.xr[i] = q < (p.C >> 2) ? reinterpret_cast<const float4 *>(row)[q] : make_float4(0, 0, 0, 0);
        }
    }
}

template <int K, bool VEC4>
__device__ __forceinline__ void blank_row_finish(const BlankParams &p, const BlankRow<K> &r, float *occ, float *gam)
{
    const int lane = lane_id(), s0 = lane * K;
    float *g = p.grad + ((int64_t)r.t * p.B + r.b) * p.C;
    if (!r.live) {
        if (VEC4) {
            for (int q = lane; q < (p.C >> 2); q += kWave) stream_store(reinterpret_cast<float4 *>(g) + q, make_float4(0, 0, 0, 0));
        } else {
            for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], 0.f);
        }
        return;
    }
    const int n = 2 * r.L + 1;
    const int *cls = p.cls + r.b * p.NSP, *nxt = p.nxt + r.b * p.NSP, *first = p.first + r.b * p.NSP;
    float v[K];
    float m = kNegB;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? r.al[k] + r.be[k] - r.em[k] : kNegB;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float ssum = 0.f, blank_part = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? __builtin_amdgcn_exp2f(v[k] - m) : 0.f;     // lattice is in log2 units
        ssum += v[k];
        if (((s0 + k) & 1) == 0) blank_part += v[k];
    }
    ssum = wave_sum(ssum);
    blank_part = wave_sum(blank_part);
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int k = 0; k < K; ++k) gam[s0 + k] = v[k] * inv;            // wave-local LDS, in order
    // occupancy per class: blank from the reduction, labels folded along the repeat chain
    if (lane == 0) occ[p.blank] = blank_part * inv;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        if (first[s]) {                                       // label states only; repeats are chained
            float tot = gam[s];
            for (int q = nxt[s]; q >= 0; q = nxt[q]) tot += gam[q];
            occ[cls[s]] = tot;
        }
    }
    const float gs = p.grad_scale / (float)(r.L > 1 ? r.L : 1);
    if (VEC4) {
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) {
            const int q = lane + kWave * i;
            if (q < (p.C >> 2)) {
                const float4 o = reinterpret_cast<const float4 *>(occ)[q];
                float4 out;
                out.x = (fast_exp(r.xr[i].x) - o.x) * gs;
                out.y = (fast_exp(r.xr[i].y) - o.y) * gs;
                out.z = (fast_exp(r.xr[i].z) - o.z) * gs;
                out.w = (fast_exp(r.xr[i].w) - o.w) * gs;
                stream_store(reinterpret_cast<float4 *>(g) + q, out);
            }
        }
    } else {
        const float *row = p.lp + (int64_t)r.t * p.st + (int64_t)r.b * p.sb;
        for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], (fast_exp(row[c]) - occ[c]) * gs);
    }
    // un-set only what this row touched
    if (lane == 0) occ[p.blank] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        if ((s & 1) && s < n) occ[cls[s]] = 0.f;
    }
}

// VEC4: C % 4 == 0 and 16-byte aligned rows -> the dense part moves float4 per lane (4x fewer
// memory instructions).  Rows are double-buffered: the loads of a wave's NEXT row (lattice
// triples + the whole log-prob row, kMaxV4 float4 per lane cover C <= 1024) are in flight while
// the current row is reduced and written.
template <int K, bool VEC4>
__global__ __launch_bounds__(kGradWaves * kWave) void blank_grad_kernel(BlankParams p, int total_rows)
{
    extern __shared__ float4 s_buf4[];                       // per wave: occ[C4] + gam[NSP]
    const int w = wave_id(), lane = lane_id();
    const int C4 = (p.C + 3) & ~3;
    float *occ = reinterpret_cast<float *>(s_buf4) + (size_t)w * (C4 + p.NSP);
    float *gam = occ + C4;
    for (int c = lane; c < C4; c += kWave) occ[c] = 0.f;
    const int stride = gridDim.x * kGradWaves;
    int idx = blockIdx.x * kGradWaves + w;
    if (idx >= total_rows) return;
    BlankRow<K> ra, rb;
    blank_row_load<K, VEC4>(p, idx, ra);
    for (; idx < total_rows; idx += 2 * stride) {
        const bool has_b = idx + stride < total_rows;        // wave-uniform
        if (has_b) blank_row_load<K, VEC4>(p, idx + stride, rb);
        blank_row_finish<K, VEC4>(p, ra, occ, gam);
        if (idx + 2 * stride < total_rows) blank_row_load<K, VEC4>(p, idx + 2 * stride, ra);
        if (has_b) blank_row_finish<K, VEC4>(p, rb, occ, gam);
    }
}

template <int K>
static int run_blank(BlankParams &p, hipStream_t s)
{
    p.NSP = kWave * K;
    const size_t lattice = (size_t)p.B * p.T * p.NSP;
    float *base = reinterpret_cast<float *>(reinterpret_cast<char *>(p.counter) + 256);
    p.em = base;
    p.al = base + lattice;
    p.be = base + 2 * lattice;
    p.cls = reinterpret_cast<int *>(base + 3 * lattice);
    p.nxt = p.cls + (size_t)p.B * p.NSP;
    p.first = p.nxt + (size_t)p.B * p.NSP;
    const int rows_per_block = 8;
    int rc = launch<blank_gather_kernel>(dim3((p.T + rows_per_block - 1) / rows_per_block, p.B), dim3(256),
                                         p.NSP * sizeof(int), s, p, rows_per_block);
    if (rc) return rc;
    rc = launch<blank_chain_kernel<K>>(dim3(p.B), dim3(128), 0, s, p);
    if (rc || !p.grad) return rc;
    const int total = p.T * p.B;
    int blocks = (total + kGradWaves - 1) / kGradWaves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = (size_t)kGradWaves * (((p.C + 3) & ~3) + p.NSP) * sizeof(float);
    const bool vec4 = (p.C % 4 == 0) && p.C <= 4 * kWave * kMaxV4 && (p.st % 4 == 0) && (p.sb % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(p.lp) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.grad) % 16 == 0);
    if (vec4) return launch<blank_grad_kernel<K, true>>(dim3(blocks), dim3(kGradWaves * kWave), lds, s, p, total);
    return launch<blank_grad_kernel<K, false>>(dim3(blocks), dim3(kGradWaves * kWave), lds, s, p, total);
}

}  // namespace ctc

using namespace ctc;

extern "C" int ctc_amd_blank_loss_grad(const float *log_probs, int64_t stride_t, int64_t stride_b,
                                       const void *targets, int targets_i64,
                                       const int64_t *in_len, const int64_t *tgt_len,
                                       int T, int B, int C, int S, int blank,
                                       float loss_scale, float grad_scale,
                                       float *nll, float *loss, float *grad,
                                       void *workspace, void *stream)
{
    if (!log_probs || !targets || !in_len || !tgt_len || !nll || !loss || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1 || blank < 0 || blank >= C) return CTC_AMD_ERR_BAD_ARGUMENT;
    const int ns = 2 * S + 1;
    if (ns > kWave * 8) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;                 // S <= 255
    if ((size_t)kGradWaves * (C + 4 + kWave * 8) * sizeof(float) > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    BlankParams p;
    p.lp = log_probs; p.st = stride_t; p.sb = stride_b;
    p.tgt = targets; p.tgt64 = targets_i64;
    p.in_len = in_len; p.tgt_len = tgt_len;
    p.T = T; p.B = B; p.C = C; p.S = S; p.blank = blank;
    p.loss_scale = loss_scale; p.grad_scale = grad_scale;
    p.nll = nll; p.loss = loss; p.grad = grad;
    p.counter = static_cast<unsigned *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ns <= kWave * 2) return run_blank<2>(p, s);
    if (ns <= kWave * 4) return run_blank<4>(p, s);
    return run_blank<8>(p, s);
}
