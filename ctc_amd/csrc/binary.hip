// Fused no-blank BINARY (multi-label sigmoid) CTC loss + input gradient for gfx950.
//
// Replaces NoBlankBinaryCTC.forward (NoBlankBinaryCTC.py:139-151) and the autograd
// backward over it.  Same lattice as the no-blank loss (lattice.hpp); what differs is
// the emission: cell (t,l) costs -nn.BCELoss()(sigmoid(x[t,b,:]), y[b,l,:])
// (NoBlankBinaryCTC.py:112,:88; mean over C, logs clamped at -100 as torch does):
//
//   lp_c = max(log p_c, -100), lq_c = max(log(1 - p_c), -100), p = sigmoid(x)   (:146)
//   e[t,l] = (1/C) * ( sum_c y[l,c] * (lp_c - lq_c)  +  sum_c lq_c )
//
// i.e. one [T x C] . [C x S] contraction per sample, and for the gradient the
// transposed one:  grad[t,c] = scale/C * ( p_c * sum_l gamma_t(l) - sum_l gamma_t(l) y[l,c] )
// times p(1-p)/max(p(1-p),1e-12) (torch's BCELoss backward floors the denominator).
//
// One 16-wave workgroup per sample: y[b] is staged once in LDS; every wave takes rows
// t = w, w+16, ...: elementwise lp/lq into a wave-private LDS row, then S dot products
// (lanes over c, DPP wave reduction).  Chains and posteriors as in noblank.hip.  The
// gradient pass contracts gamma_t with the staged y per element and re-reads x (L2).
#include "lattice.hpp"
#include "launch.hpp"

namespace ctc {

struct BinaryParams {
    const float *x;
    int64_t st, sb;
    const float *y;
    const int64_t *in_len, *tgt_len;
    int T, B, C, S, SP, CP;      // CP: padded row pitch of the staged y / row buffers
    float loss_scale, grad_scale;
    float *nll, *loss, *grad;
    unsigned *counter;
};

constexpr int kBinThreads = 1024;
constexpr int kBinWaves = kBinThreads / kWave;

struct BinarySmem {
    float *em, *al, *be, *dummy, *ys, *drow;
    __device__ BinarySmem(float *base, int T, int SP, int S, int CP)
    {
        em = base + kPrefetch * SP;
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        dummy = be + (size_t)T * SP;
        ys = dummy + 8;
        drow = ys + (size_t)S * CP;
    }
};

static size_t binary_smem_bytes(int T, int SP, int S, int CP)
{
    return ((size_t)(3 * T + 2 * kPrefetch) * SP + 8 + (size_t)S * CP + (size_t)kBinWaves * CP) * 4;
}

// p = sigmoid(x) in fp32, then the two clamped logs exactly as nn.BCELoss sees them
__device__ __forceinline__ void bce_logs(float x, float &p, float &lp, float &lq)
{
    p = 1.0f / (1.0f + expf(-x));
    lp = fmaxf(logf(p), -100.0f);
    lq = fmaxf(logf(1.0f - p), -100.0f);
}

template <int K>
__global__ __launch_bounds__(kBinThreads) void binary_fused_kernel(BinaryParams p)
{
    extern __shared__ float4 smem_raw[];
    const BinarySmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.S, p.CP);
    const int b = blockIdx.x, tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;

    // stage y[b] (S x C, contiguous) into LDS with pitch CP
    const float *yb = p.y + (int64_t)b * p.S * p.C;
    for (int i = tid; i < p.S * p.C; i += kBinThreads) {
        const int l = i / p.C, c = i - l * p.C;
        sm.ys[l * p.CP + c] = yb[i];
    }
    if (tid < 8) sm.dummy[tid] = 0.f;
    for (int i = tid; i < kPrefetch * p.SP; i += kBinThreads) {
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }
    __syncthreads();

    // P1: emissions
    float *dw = sm.drow + w * p.CP;
    const float invC = 1.0f / (float)p.C;
    for (int t = w; t < Tb; t += kBinWaves) {
        const float *row = p.x + (int64_t)t * p.st + (int64_t)b * p.sb;
        float q = 0.f;
        for (int c = lane; c < p.C; c += kWave) {
            float pr, lp, lq;
            bce_logs(row[c], pr, lp, lq);
            dw[c] = lp - lq;
            q += lq;
        }
        q = wave_sum(q);
        for (int l = 0; l < p.SP; ++l) {
            float s = 0.f;
            if (l < L) {                                       // wave-uniform
                const float *yl = sm.ys + l * p.CP;
                for (int c = lane; c < p.C; c += kWave) s = __builtin_fmaf(dw[c], yl[c], s);
                s = wave_sum(s);
            }
            if (lane == 0) sm.em[t * p.SP + l] = (l < L) ? (s + q) * invC : kNeg;
        }
    }
    __syncthreads();

    // P2: alpha / beta' chains
    if (Tb > 0) {
        const bool rot = p.SP <= 63 * K;
        if (w == 0) {
            if (rot) lattice_chain<K, true, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP);
            else lattice_chain<K, true, false>(sm.em, sm.al, sm.dummy, Tb, L, p.SP);
        } else if (w == 1 && p.grad) {
            if (rot) lattice_chain<K, false, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP);
            else lattice_chain<K, false, false>(sm.em, sm.be, sm.dummy, Tb, L, p.SP);
        }
    }
    __syncthreads();

    const float nll = ok ? -sm.al[(Tb - 1) * p.SP + (L - 1)] : -kNeg;
    if (w == kBinWaves - 1)
        publish_and_reduce(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter,
                           [](float v, int) { return v; });
    if (!p.grad) return;

    // P3: posteriors of this wave's rows, then the gradient rows
    const bool feasible = ok && nll < kInfeasible;
    const int Tlive = feasible ? Tb : 0;
    const int G = posterior_group(p.SP), per = kWave / G, sub = lane / G;
    const float gs = p.grad_scale * invC;
    for (int t0 = w * per; t0 < p.T; t0 += kBinWaves * per) {
        if (t0 < Tlive)
            posterior_row<false>(sm.al, sm.be, sm.em, nullptr, nullptr, t0 + sub, t0 + sub < Tlive, L, p.SP, G);
        for (int r = 0; r < per; ++r) {
            const int t = t0 + r;
            if (t >= p.T) break;
            float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
            if (t < Tlive) {
                const float *row = p.x + (int64_t)t * p.st + (int64_t)b * p.sb;
                const float *gam = sm.be + t * p.SP;
                float tot = 0.f;
                for (int l = 0; l < L; ++l) tot += gam[l];
                for (int c = lane; c < p.C; c += kWave) {
                    const float xv = row[c];
                    const float pr = 1.0f / (1.0f + expf(-xv));
                    float occ = 0.f;
                    for (int l = 0; l < L; ++l) occ = __builtin_fmaf(gam[l], sm.ys[l * p.CP + c], occ);
                    const float pq = pr * (1.0f - pr);
                    g[c] = gs * (pr * tot - occ) * (pq / fmaxf(pq, 1e-12f));
                }
            } else {
                for (int c = lane; c < p.C; c += kWave) g[c] = 0.f;
            }
        }
    }
}

}  // namespace ctc

using namespace ctc;

extern "C" int ctc_amd_binary_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                                        const float *y,
                                        const int64_t *in_len, const int64_t *tgt_len,
                                        int T, int B, int C, int S,
                                        float loss_scale, float grad_scale,
                                        float *nll, float *loss, float *grad,
                                        void *workspace, void *stream)
{
    if (!x || !y || !in_len || !tgt_len || !nll || !loss || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    int K = 1;
    while (K <= 4 && S > kWave * K) K *= 2;
    if (K > 4) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    BinaryParams p;
    p.x = x; p.st = stride_t; p.sb = stride_b; p.y = y;
    p.in_len = in_len; p.tgt_len = tgt_len;
    p.T = T; p.B = B; p.C = C; p.S = S;
    p.SP = (S + K - 1) / K * K;
    p.CP = C | 1;                                            // odd pitch: conflict-free column walks
    p.loss_scale = loss_scale; p.grad_scale = grad_scale;
    p.nll = nll; p.loss = loss; p.grad = grad;
    p.counter = static_cast<unsigned *>(workspace);
    const size_t smem = binary_smem_bytes(T, p.SP, S, p.CP);
    if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 grid(B), block(kBinThreads);
    switch (K) {
        case 1: return launch<binary_fused_kernel<1>>(grid, block, smem, s, p);
        case 2: return launch<binary_fused_kernel<2>>(grid, block, smem, s, p);
        default: return launch<binary_fused_kernel<4>>(grid, block, smem, s, p);
    }
}
