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
#include <cstdlib>
#include <type_traits>

#include "lattice.hpp"
#include "launch.hpp"

namespace ctc {

struct BinaryParams {
    const float *x;
    int64_t st, sb;
    const float *y;
    const int64_t *in_len, *tgt_len;
    int T, B, C, S, SP, CP;      // CP: padded row pitch of the staged y / row buffers
    int stop;                    // diagnostics: < 0 selects the wave that stamps phase boundaries
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

// The same numbers with fewer instructions (the logs of the MFMA kernel are pure VALU throughput):
//   * exp(-x): the library expf's own algorithm (product split into a round-to-nearest integer and
//     a compensated fraction, v_exp_f32, v_ldexp_f32) without its range clamps -- beyond them the
//     plain sequence already yields inf / 0 / a denormal, which is what the clamps return; p
//     itself must be the library's value to the last bit (where 1 - p cancels, the reference's
//     log(1 - p) depends on it);
//   * log p = x - log1p(e^x) equals x to fp32 precision for x < -30 (and v_log_f32 does not take
//     the denormal p of x < -87): a select instead of a divergent branch; log(1 - p) on the raw
//     v_log_f32 (1 - p is either 0 or >= 2^-24), which is good to ~1e-7 relative.
__device__ __forceinline__ void bce_logs_fast(float x, float &p, float &lp, float &lq)
{
    const float t = x * -kLog2e;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(x, -kLog2e, -t);                // rounding error of the product
    r = __builtin_fmaf(x, -1.92596299e-8f, r);               // + x * (low part of -log2 e)
    const float e = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f((t - n) + r), (int)n);
    p = __builtin_amdgcn_rcpf(1.0f + e);
    const float one_m_p = 1.0f - p;
    const float lp_raw = __builtin_amdgcn_logf(p) * kLn2;
    lp = x < -30.0f ? fmaxf(x, -100.0f) : lp_raw;
    lq = one_m_p > 0.0f ? fmaxf(__builtin_amdgcn_logf(one_m_p) * kLn2, -100.0f) : -100.0f;
}

template <int K>
__global__ __launch_bounds__(kBinThreads) void binary_fused_kernel(BinaryParams p)
{
    extern __shared__ float4 smem_raw[];
    const BinarySmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.S, p.CP);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
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
        publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
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
                    stream_store(&g[c], gs * (pr * tot - occ) * (pq / fmaxf(pq, 1e-12f)));
                }
            } else {
                for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], 0.f);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// MFMA variant (the fast path when the images fit in LDS).  Both contractions are real
// [T x C].[C x S] GEMMs (0.95 MFLOP per sample each way), so this is the one place on the
// CTC path where the matrix cores are the right tool: v_mfma_f32_16x16x4_f32 is exact fp32
// (a k-ordered fmaf chain) at the fp32 vector rate, needs no cross-lane reduction and
// leaves the VALU to the elementwise work.
//
//   P1a every wave: its kBinRows rows of x (kept in registers for P3c) -> d = lp - lq into an
//       LDS image D[t][c] (pitch PD), Q[t] = sum_c lq by a DPP reduction
//   P1b E = D . Y^T on MFMA: A fragment = D[16m + (lane&15)][4k + (lane>>4)],
//       B fragment = Y[16n + (lane&15)][4k + (lane>>4)]; PD = 2 (mod 32) makes both
//       ds_read_b32 patterns conflict-free; e[t,l] = (E + Q[t]) / C
//   P2  chains (lattice.hpp)
//   P3a posteriors per row (gamma, 0 beyond L)      P3b G = gamma . Y on MFMA into the D image
//   P3c grad[t,c] = scale/C * (sigmoid(x) - G[t,c]) * p(1-p)/max(p(1-p),1e-12) from the
//       resident rows, coalesced stores
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBinRows = 160 / kBinWaves;                    // rows resident per wave (T <= 160)

struct BinaryMfmaSmem {
    float *em, *al, *be, *dummy, *q, *ys, *dimg;
    int *prog;                                               // [2] completed steps of the alpha / beta' scans
    __device__ BinaryMfmaSmem(float *base, int T, int Tpad, int SP, int PD)
    {
        em = base + kPrefetch * SP;
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        dummy = be + (size_t)T * SP;
        prog = reinterpret_cast<int *>(dummy + 4);           // (dummy[0..3]: idle-lane slots)
        q = dummy + 8;
        ys = q + Tpad;
        dimg = ys + (size_t)SP * PD;
    }
};

static size_t binary_mfma_smem_bytes(int T, int Tpad, int SP, int PD)
{
    return ((size_t)(3 * T + 2 * kPrefetch) * SP + 8 + Tpad + (size_t)SP * PD + (size_t)Tpad * PD) * 4;
}

template <int K, int CH>
__global__ __launch_bounds__(kBinThreads) void binary_mfma_kernel(BinaryParams p, int Tpad, int PD)
{
    extern __shared__ float4 smem_raw[];
    const BinaryMfmaSmem sm(reinterpret_cast<float *>(smem_raw), p.T, Tpad, p.SP, PD);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    stamp(p, 0);
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];

    // rows of this wave, unconditional clamped loads (stay resident until P3c)
    float v[kBinRows][CH];
#pragma unroll
    for (int r = 0; r < kBinRows; ++r) {
        const int t = w * kBinRows + r;
        const float *row = p.x + (int64_t)(t < p.T ? t : p.T - 1) * p.st + (int64_t)b * p.sb;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = lane + 64 * j;
            v[r][j] = row[c < p.C ? c : p.C - 1];
        }
    }
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;

    // y[b] -> LDS image [SP][PD], zero beyond S rows / C columns (MFMA K and N padding)
    const float *yb = p.y + (int64_t)b * p.S * p.C;
    for (int i = tid; i < p.SP * PD; i += kBinThreads) {
        const int l = i / PD, c = i - l * PD;
        sm.ys[i] = (l < p.S && c < p.C) ? yb[l * p.C + c] : 0.f;
    }
    if (tid < 8) sm.dummy[tid] = 0.f;                        // (also zeroes the two progress counters)
    for (int i = tid; i < kPrefetch * p.SP; i += kBinThreads) {
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }

    stamp(p, 1);
    // P1a: elementwise BCE logs -> D image, Q
#pragma unroll
    for (int r = 0; r < kBinRows; ++r) {
        const int t = w * kBinRows + r;
        if (t >= Tpad) break;                                // wave-uniform
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = lane + 64 * j;
            float pr, lp, lq;
            bce_logs_fast(v[r][j], pr, lp, lq);
            v[r][j] = pr;                                    // the gradient needs sigmoid(x), not x
            const bool in = c < p.C && t < Tb;
            if (c < PD) sm.dimg[t * PD + c] = in ? lp - lq : 0.f;    // zero K padding / dead rows
            q += in ? lq : 0.f;
        }
        q = wave_sum(q);
        if (lane == 0) sm.q[t] = q;
    }
    stamp(p, 2);
    __syncthreads();

    // P1b: E = D . Y^T (16x16 tiles, K = C in steps of 4)
    const int MT = Tpad >> 4, NT = (p.SP + 15) >> 4, KT = (p.C + 3) >> 2;
    const float invC = 1.0f / (float)p.C;
    const int fr = lane & 15, fq = lane >> 4;
    for (int job = w; job < MT * NT; job += kBinWaves) {
        const int m = job / NT, n = job - m * NT;
        const int lrow = 16 * n + fr;
        const float *ap = sm.dimg + (16 * m + fr) * PD + fq;
        const float *bp = sm.ys + (lrow < p.SP ? lrow : p.SP - 1) * PD + fq;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < KT; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[4 * kk], acc, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 16 * m + 4 * fq + j;
            if (t < Tb && lrow < p.SP) sm.em[t * p.SP + lrow] = lrow < L ? (acc[j] + sm.q[t]) * invC : kNeg;
        }
    }
    stamp(p, 3);
    __syncthreads();
    stamp(p, 4);

    // P2: alpha / beta' chains (waves 0 and 1).  S <= 64: they publish their progress and NO
    // barrier follows -- every wave then finishes its own rows as soon as both scans have
    // passed them (middle of the sequence first), overlapping the gradient with the scans.
    const float gs = p.grad_scale * invC;
    if (Tb > 0) {
        const bool rot = p.SP <= 63 * K;
        int *pa = K == 1 ? sm.prog : nullptr, *pb = K == 1 ? sm.prog + 1 : nullptr;
        if (w == 0) {
            if (rot) lattice_chain<K, true, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, pa);
            else lattice_chain<K, true, false>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, pa);
        } else if (w == 1 && p.grad) {
            if (rot) lattice_chain<K, false, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, pb);
            else lattice_chain<K, false, false>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, pb);
        }
    }
    stamp(p, 5);
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    bool starved = false;                                    // a bounded wait ran out: NaN outputs, status raised
    if (K != 1) __syncthreads();
    if (w == kBinWaves - 1 || !p.grad) {
        // nll / batch mean by the last wave (it owns the fewest rows): wait for the alpha scan
        if (K == 1 && Tb > 0) {
            int spins = 0;
            while (*(lds_cvint *)sm.prog < Tb) {
                if (++spins >= (1 << 20)) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();
        }
        if (w == kBinWaves - 1) {
            float nll = ok ? -sm.al[(Tb - 1) * p.SP + (L - 1)] : -kNeg;
            if (starved) {                                   // a bounded wait ran out: NaN, not a plausible number
                nll = __builtin_nanf("");
                raise_status(p.counter, kStatusBinaryStarved);
            }
            publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        }
        if (!p.grad) return;
    }
    const int Tlive = Tb;                                    // ok <=> an alignment exists (L_b <= T_b)

    if (K == 1) {
        // P3 (S <= 64): every wave finishes its OWN rows, four at a time, with no further barrier.
        //   gamma rows by posterior_rows4 (wave-local LDS), then G = gamma . Y on
        //   v_mfma_f32_4x4x1 (16 blocks of 4x4, K = 1 per instruction): block = lane>>2 and
        //   column-in-block = lane&3 make the output column equal the lane and put the four rows
        //   in the four accumulator registers -- exactly the layout of the resident rows, so
        //   the gradient needs no LDS transpose.  A = gamma[row lane&3][l], B = Y[l][lane + 64 j].
        auto group = [&](auto R0) {
            constexpr int r0 = decltype(R0)::value;
            int tt[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = w * kBinRows + r0 + i;
                tt[i] = (r0 + i < kBinRows && t < Tlive) ? t : -1;
            }
            const int t_first = w * kBinRows + r0;
            if (t_first >= p.T) return;                      // wave-uniform
            f32x4 acc[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t_first < Tlive) {                           // wave-uniform: at least one live row
                int t_hi = t_first;                          // the scans must have passed rows t_first..t_hi
#pragma unroll
                for (int i = 0; i < 4; ++i) t_hi = tt[i] >= 0 ? tt[i] : t_hi;
                int spins = 0;
                while (*(lds_cvint *)sm.prog < t_hi + 1 || *(lds_cvint *)(sm.prog + 1) < Tlive - t_first) {
                    if (++spins >= (1 << 20)) { starved = true; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                lds_order();
                if (starved) raise_status(p.counter, kStatusBinaryStarved);
                posterior_rows4(sm.al, sm.be, sm.em, tt, L, p.SP, 1.0f);
                const int ti = tt[lane & 3];
                const float *arow = sm.be + (ti >= 0 ? ti : 0) * p.SP;
                const float amask = ti >= 0 ? 1.f : 0.f;
                for (int l0 = 0; l0 < L; l0 += 4) {          // SP % 4 == 0; gamma and Y rows beyond L / S are 0
                    float fa[4], fy[4][CH];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {            // loads first: LDS latency once per batch
                        fa[i] = arow[l0 + i] * amask;
#pragma unroll
                        for (int j = 0; j < CH; ++j) fy[i][j] = sm.ys[(l0 + i) * PD + lane + 64 * j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < CH; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa[i], fy[i][j], acc[j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = w * kBinRows + r0 + i;
                if (r0 + i >= kBinRows || t >= p.T) break;   // wave-uniform
                float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
                const bool live = t < Tlive;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    if (c < p.C) {
                        const float pr = v[r0 + i][j];       // sigmoid(x), kept from P1a
                        const float pq = pr * (1.0f - pr);
                        const float gv = live ? gs * (pr - acc[j][i]) * (pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f))) : 0.f;
                        stream_store(&g[c], starved ? __builtin_nanf("") : gv);
                    }
                }
            }
        };
        // rows nearest the middle of the sequence are ready first: walk this wave's groups in that order
        static_assert(kBinRows > 8 && kBinRows <= 12, "three groups of four rows per wave");
        using std::integral_constant;
        if (w * kBinRows * 2 + kBinRows < Tlive) {           // wave in the first half: highest rows first
            group(integral_constant<int, 8>{});
            group(integral_constant<int, 4>{});
            group(integral_constant<int, 0>{});
        } else {
            group(integral_constant<int, 0>{});
            group(integral_constant<int, 4>{});
            group(integral_constant<int, 8>{});
        }
        stamp(p, 7);
        return;
    }

    // K > 1 (S > 64): posteriors row by row, G on 16x16 MFMA tiles through the D image
    {
        const int G = posterior_group(p.SP), per = kWave / G, sub = lane / G;
        for (int t0 = w * per; t0 < Tlive; t0 += kBinWaves * per)
            posterior_row<false>(sm.al, sm.be, sm.em, nullptr, nullptr, t0 + sub, t0 + sub < Tlive, L, p.SP, G);
    }
    __syncthreads();

    // P3b: G = gamma . Y (M = t, N = c, K = l) into the D image
    const int NC = (p.C + 15) >> 4, KL = p.SP >> 2;
    for (int job = w; job < MT * 2; job += kBinWaves) {
        const int m = job >> 1, half = job & 1;
        const int n_lo = half ? (NC + 1) / 2 : 0, n_hi = half ? NC : (NC + 1) / 2;
        const float *ap = sm.be + (16 * m + fr) * p.SP + fq;
        for (int n = n_lo; n < n_hi; ++n) {
            const float *bp = sm.ys + fq * PD + 16 * n + fr;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < KL; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[4 * kk * PD], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) sm.dimg[(16 * m + 4 * fq + j) * PD + 16 * n + fr] = acc[j];
        }
    }
    __syncthreads();

    // P3c: gradient rows from the resident registers
#pragma unroll
    for (int r = 0; r < kBinRows; ++r) {
        const int t = w * kBinRows + r;
        if (t >= p.T) break;                                 // wave-uniform
        float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
        const bool live = t < Tlive;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = lane + 64 * j;
            if (c < p.C) {
                const float pr = v[r][j];                    // sigmoid(x), kept from P1a
                const float pq = pr * (1.0f - pr);
                const float occ = sm.dimg[t * PD + c];
                stream_store(&g[c], live ? gs * (pr - occ) * (pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f))) : 0.f);
            }
        }
    }
}

template <int K>
static int launch_binary_mfma(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int Tpad, int PD)
{
    const dim3 grid(p.B), block(kBinThreads);
    switch (ch) {
        case 1: return launch<binary_mfma_kernel<K, 1>>(grid, block, smem, s, p, Tpad, PD);
        case 2: return launch<binary_mfma_kernel<K, 2>>(grid, block, smem, s, p, Tpad, PD);
        case 3: return launch<binary_mfma_kernel<K, 3>>(grid, block, smem, s, p, Tpad, PD);
        default: return launch<binary_mfma_kernel<K, 4>>(grid, block, smem, s, p, Tpad, PD);
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
    static const int debug_stop = diag_env("CTC_AMD_DEBUG_STOP");
    static const bool binary_valu = diag_env("CTC_AMD_BINARY_VALU") != 0;
    p.stop = debug_stop < 0 ? debug_stop : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // fast path: MFMA contractions, rows resident (T <= 160, C <= 256, images fit in LDS)
    if (T <= kBinRows * kBinWaves && C <= 256 && !binary_valu) {
        BinaryParams q = p;
        q.SP = (p.SP + 3) / 4 * 4;                           // K padding of the gamma . Y product
        if (q.SP % K) q.SP = (q.SP + 4 * K - 1) / (4 * K) * (4 * K);
        const int Tpad = (T + 15) / 16 * 16;
        int PD = (C + 3) / 4 * 4 + 2;                        // >= C (+K padding), PD = 2 (mod 32):
        while (PD % 32 != 2) PD += 2;                        // conflict-free MFMA fragment reads
        const size_t need = binary_mfma_smem_bytes(T, Tpad, q.SP, PD);
        if (need <= kMaxLds) {
            const int ch = (C + kWave - 1) / kWave;
            switch (K) {
                case 1: return launch_binary_mfma<1>(ch, need, s, q, Tpad, PD);
                case 2: return launch_binary_mfma<2>(ch, need, s, q, Tpad, PD);
                default: return launch_binary_mfma<4>(ch, need, s, q, Tpad, PD);
            }
        }
    }
    const size_t smem = binary_smem_bytes(T, p.SP, S, p.CP);
    if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    const dim3 grid(B), block(kBinThreads);
    switch (K) {
        case 1: return launch<binary_fused_kernel<1>>(grid, block, smem, s, p);
        case 2: return launch<binary_fused_kernel<2>>(grid, block, smem, s, p);
        default: return launch<binary_fused_kernel<4>>(grid, block, smem, s, p);
    }
}
