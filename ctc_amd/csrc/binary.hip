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
    float *gamma;                // optional [B][T][S] posteriors output (ctc_amd_binary_posteriors; pipelined kernel only)
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

// The same three quantities where no clamp can fire and 1 - p keeps its bits: |x| < 6 on every lane of the
// wave (p in [2.5e-3, 1 - 2.5e-3]).  Three transcendentals and five plain operations instead of four and
// thirty (the selects, the exact exponential and the second logarithm of bce_logs_fast exist for the tails):
//     e = exp(-x), p = 1/(1 + e), log p = -log(1 + e), log(1 - p) = log p - x.
// log(1 - p) is exact here where the reference rounds 1 - fl(p) first: they differ by <= 2.4e-5 at |x| = 6,
// 1e-6 of an emission after the mean over C -- the exact (float64) value is the closer one of the two.
// gradient store: write-through while logits + gradient fit the memory-side cache, non-temporal beyond (common.hpp)
template <bool WT>
__device__ __forceinline__ void bin_store(float *p, float v)
{
    if (WT) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else stream_store(p, v);
}

constexpr float kBceFastAbs = 6.0f;
__device__ __forceinline__ void bce_logs_small(float x, float &p, float &lp, float &lq)
{
    const float s = 1.0f + __builtin_amdgcn_exp2f(x * -kLog2e);
    p = __builtin_amdgcn_rcpf(s);
    lp = __builtin_amdgcn_logf(s) * -kLn2;
    lq = lp - x;
}
// wave-uniform: every lane's |x| is small (NaN fails the test and takes the careful path)
__device__ __forceinline__ bool bce_all_small(float x)
{
    return __builtin_amdgcn_ballot_w64(!(fabsf(x) < kBceFastAbs)) == 0;
}

template <int K>
__global__ __launch_bounds__(kBinThreads) void binary_fused_kernel(BinaryParams p)
{
    extern __shared__ float4 smem_raw[];
    const BinarySmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.S, p.CP);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64) note_arrival(p.counter, b);                  // (arrivals word, common.hpp)
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

template <int K, int CH, bool WT>
__global__ __launch_bounds__(kBinThreads) void binary_mfma_kernel(BinaryParams p, int Tpad, int PD)
{
    extern __shared__ float4 smem_raw[];
    const BinaryMfmaSmem sm(reinterpret_cast<float *>(smem_raw), p.T, Tpad, p.SP, PD);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64) note_arrival(p.counter, b);                  // (arrivals word, common.hpp)
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
        // (no early exit for rows beyond Tpad: every path must consume the loads above, or the compiler guards the
        // first later use of v[][] with a vmcnt wait -- which, in P3, waits for the gradient stores in flight)
        const int t = w * kBinRows + r;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int c = lane + 64 * j;
            float pr, lp, lq, d;
            const float xv = v[r][j];
            if (bce_all_small(xv)) {                         // (wave-uniform) the usual case: no tails in this chunk
                bce_logs_small(xv, pr, lp, lq);
                d = xv;                                      // lp - lq
            } else {
                bce_logs_fast(xv, pr, lp, lq);
                d = lp - lq;
            }
            v[r][j] = pr;                                    // the gradient needs sigmoid(x), not x
            const bool in = c < p.C && t < Tb;
            if (c < PD && t < Tpad) sm.dimg[t * PD + c] = in ? d : 0.f;  // zero K padding / dead rows
            q += in ? lq : 0.f;
        }
        q = wave_sum(q);
        if (lane == 0 && t < Tpad) sm.q[t] = q;
    }
    stamp(p, 2);
    __syncthreads();
    stamp(p, 8);

    // P1b: E = D . Y^T (16x16 tiles, K = C in steps of 4)
    const int MT = Tpad >> 4, NT = (p.SP + 15) >> 4, KT = (p.C + 3) >> 2;
    const float invC = 1.0f / (float)p.C;
    const int fr = lane & 15, fq = lane >> 4;
    for (int job = w; job < MT * NT; job += kBinWaves) {
        const int m = job / NT, n = job - m * NT;
        const int lrow = 16 * n + fr;
        const float *ap = sm.dimg + (16 * m + fr) * PD + fq;
        const float *bp = sm.ys + (lrow < p.SP ? lrow : p.SP - 1) * PD + fq;
        // K in blocks of eight steps: sixteen fragment reads in flight, then eight MFMAs (with a runtime trip
        // count the compiler serialises read -> wait -> MFMA: an LDS round trip per step)
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int kk = 0;
        for (; kk + 8 <= KT; kk += 8) {
            float fa[8], fb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { fa[j] = ap[4 * (kk + j)]; fb[j] = bp[4 * (kk + j)]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], acc, 0, 0, 0);
        }
        for (; kk < KT; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * kk], bp[4 * kk], acc, 0, 0, 0);
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
                if (r0 == 8) stamp(p, 9);
                posterior_rows4(sm.al, sm.be, sm.em, tt, L, p.SP, 1.0f);
                if (r0 == 8) stamp(p, 10);
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
            if (r0 == 8) stamp(p, 11);
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
                        // (the factor p(1-p)/max(p(1-p), 1e-12) is 1 unless |x| > 27: skipped when no lane needs it)
                        float gv = gs * (pr - acc[j][i]);
                        if (__builtin_amdgcn_ballot_w64(!(pq >= 1e-12f)) != 0) gv *= pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f));
                        if (!live) gv = 0.f;
                        bin_store<WT>(&g[c], starved ? __builtin_nanf("") : gv);
                    }
                }
            }
        };
        stamp(p, 6);
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
                bin_store<WT>(&g[c], live ? gs * (pr - occ) * (pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f))) : 0.f);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// Pipelined variant (S <= 64, T <= 168; the default when its images fit in LDS).  Same arithmetic as
// binary_mfma_kernel<1>; what changes is who waits for whom:
//
//   waves 0 / 1   the alpha / beta' scans.  They own no rows; each starts as soon as the emission tile at ITS end
//                 of the sequence is in LDS and checks one tile flag per 16 steps.
//   waves 2..15   14 workers.  Worker u owns twelve rows: slot (g, side), g = 0..5, is the front row 14g+u or
//                 the back row T-1-(14g+u).  P1a: logs of its rows -> D image (cheap form, see bce_row below);
//                 one workgroup barrier; P1b: emission tiles E = D . Y^T (16 rows x all labels per job, 16x16x4
//                 MFMA, one job per worker, the tiles at the two ends of the sequence first and at higher wave
//                 priority) -> em, tile flag.
//                 P3: three groups of four rows per worker = the slots of rounds (4,5), (2,3), (0,1), finished
//                 in that order as both scans pass them: the rows a scan reaches last are spread over all
//                 workers instead of sitting with the owners of the sequence ends.
//   hand-offs     LDS flags only (tile[m], prog[2]) after the one barrier.
constexpr int kPipeWorkers = kBinWaves - 2;
constexpr int kPipeRounds = 6;
constexpr int kPipeMaxT = 2 * kPipeWorkers * kPipeRounds;    // 168
constexpr int kPipeMaxTiles = (kPipeMaxT + 15) / 16;         // 11 <= kPipeWorkers: one tile job per worker
static_assert(kPipeMaxTiles <= kPipeWorkers, "one emission tile job per worker");

struct BinaryPipeSmem {
    float *em, *al, *be, *dummy, *q, *zrow, *ys, *dimg;      // zrow: 64 zeros (the gamma row of an idle slot)
    int *prog;                                               // [2] completed steps of the alpha / beta' scans
    int *fail;                                               // a scan ran out of patience: outputs are NaN
    int *tile;                                               // [kPipeMaxTiles] emission tile m is in LDS
    __device__ BinaryPipeSmem(float *base, int T, int SP, int PD)
    {
        em = base + kPrefetch * SP;
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        dummy = be + (size_t)T * SP;                         // (dummy[0..3]: idle-lane slots)
        prog = reinterpret_cast<int *>(dummy + 4);
        fail = prog + 2;
        tile = prog + 4;
        q = dummy + 32;
        zrow = q + ((T + 3) & ~3);
        ys = zrow + 64;
        dimg = ys + (size_t)SP * PD;
    }
};

static size_t binary_pipe_smem_bytes(int T, int SP, int PD)
{
    return ((size_t)(3 * T + 2 * kPrefetch) * SP + 32 + ((T + 3) & ~3) + 64 + (size_t)SP * PD + (size_t)T * PD) * 4;
}

// lattice_chain<1> fed tile by tile: `ready(i)` returns once the emission rows of every step <= i are in LDS
// (steps count from the scan's own end of the sequence)
// LOG2: the lattice in units of log2 (emissions arrive multiplied by log2 e): a step is two multiplications shorter
template <bool FWD, bool ROT, bool LOG2 = false, typename Ready>
__device__ __forceinline__ void lattice_chain_fed(const float *em, float *out, float *dummy, int Tb, int L, int SP,
                                                  int *prog, Ready &&ready)
{
    const int l0 = lane_id();
    const bool act = l0 < SP;
    const int dir = FWD ? SP : -SP;
    const int t_first = FWD ? 0 : Tb - 1;
    const float *rd = act ? em + t_first * SP + l0 : em - kPrefetch * SP;
    float *wr = act ? out + t_first * SP + l0 : dummy;
    const int winc = act ? dir : 0;
    const int start = FWD ? 0 : L - 1;
    float a, ring[kPrefetch];
    auto step = [&](float e) {
        float adv;
        if (ROT) {
            adv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), FWD ? 0x13C : 0x134,
                                                                     0xf, 0xf, false));
        } else {
            adv = FWD ? wave_shr1(a, kNeg) : wave_shl1(a, kNeg);
        }
        if (LOG2) {
            const float t = __builtin_amdgcn_exp2f(-fabsf(a - adv));
            a = (__builtin_amdgcn_logf(1.0f + t) + vmax(a, adv)) + e;
        } else {
            const float t = __builtin_amdgcn_exp2f(-fabsf(a - adv) * kLog2e);
            a = __builtin_fmaf(__builtin_amdgcn_logf(1.0f + t), kLn2, vmax(a, adv)) + e;
        }
        *wr = a;
        wr += winc;
    };
    ready(kPrefetch < Tb - 1 ? kPrefetch : Tb - 1);
    a = (l0 == start) ? *rd : kNeg;                          // first step: only "stay" from the virtual start state
    *wr = a;
    wr += winc;
    rd += winc;
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) { ring[j] = *rd; rd += winc; }
    int i = 1;
    for (; i + kPrefetch <= Tb; i += kPrefetch) {
        lds_order();
        *prog = i;
        ready(i + 2 * kPrefetch - 1 < Tb - 1 ? i + 2 * kPrefetch - 1 : Tb - 1);   // this block refills the ring
#pragma unroll
        for (int j = 0; j < kPrefetch; ++j) {
            const float e = ring[j];
            ring[j] = *rd;
            rd += winc;
            step(e);
        }
    }
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j)
        if (i + j < Tb) step(ring[j]);
    lds_order();
    *prog = Tb;
}

// bce_logs_fast, returning s = 1 + exp(-x) (the reciprocal of which is the library's sigmoid to <= 1 ulp)
__device__ __forceinline__ void bce_logs_fast_s(float x, float &s, float &lp, float &lq)
{
    const float t = x * -kLog2e;
    const float n = __builtin_rintf(t);
    float r = __builtin_fmaf(x, -kLog2e, -t);
    r = __builtin_fmaf(x, -1.92596299e-8f, r);
    s = 1.0f + __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f((t - n) + r), (int)n);
    const float p = __builtin_amdgcn_rcpf(s);
    const float one_m_p = 1.0f - p;
    const float lp_raw = __builtin_amdgcn_logf(p) * kLn2;
    lp = x < -30.0f ? fmaxf(x, -100.0f) : lp_raw;
    lq = one_m_p > 0.0f ? fmaxf(__builtin_amdgcn_logf(one_m_p) * kLn2, -100.0f) : -100.0f;
}

// gradient store at row base (uniform) + lane offset + compile-time column offset: no 64-bit vector address math
template <bool WT, int OFF>
__device__ __forceinline__ void bin_store_at(float *row, unsigned voff, float v)
{
    if (WT) asm volatile("global_store_dword %0, %1, %2 offset:%3 sc1" ::"v"(voff), "v"(v), "s"(row), "n"(OFF) : "memory");
    else asm volatile("global_store_dword %0, %1, %2 offset:%3 nt" ::"v"(voff), "v"(v), "s"(row), "n"(OFF) : "memory");
}

// The row base of the stores above is an "s" operand of an inline-asm vector-memory instruction: if the compiler has just
// RELOADED it from a spill lane (v_readlane: a VALU write of an SGPR), the instruction needs 5 wait states behind that
// write, and the compiler pads them only for instructions it knows (/opt/skills/guides/cdna_hip_programming.md 5.7).  One
// statement per row, in front of its first store: it takes the base as an input, so any reload sits in front of it.
__device__ __forceinline__ void bin_row_base_settled(const float *row) { asm volatile("s_nop 4" ::"s"(row)); }

template <bool WT>
__device__ __forceinline__ void bin_store_col(float *row, unsigned voff, int j, float v)
{   // (j is a constant once the caller's loop is unrolled)
    switch (j) {
        case 0: bin_store_at<WT, 0>(row, voff, v); break;
        case 1: bin_store_at<WT, 256>(row, voff, v); break;
        case 2: bin_store_at<WT, 512>(row, voff, v); break;
        default: bin_store_at<WT, 768>(row, voff, v); break;
    }
}

template <int CH, bool WT, bool GAMMA = false>               // GAMMA: the posteriors-only instantiation (ctc_amd_binary_posteriors)
__global__ __launch_bounds__(kBinThreads) void binary_pipe_kernel(BinaryParams p, int PD)
{
    extern __shared__ float4 smem_raw[];
    const BinaryPipeSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, PD);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64) note_arrival(p.counter, b);                  // (wave 1: the beta' scan, no other vector-memory operation)
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    stamp(p, 0);
    const ScalarLengths lens(p.in_len + b, p.tgt_len + b);
    const int H = (p.T + 1) >> 1;                            // rows [0,H): front rows, [H,T): back rows
    const int NG = (H + kPipeWorkers - 1) / kPipeWorkers;    // rounds in use
    const int MT = (p.T + 15) >> 4;                          // emission tiles of 16 rows
    const int u = w - 2;                                     // worker index
    // slot (g, side) of this worker: its row, or -1 when the round does not reach that far
    auto slot_row = [&](int g, int side) {
        const int i = kPipeWorkers * g + u;
        const int t = side ? p.T - 1 - i : i;
        return (side ? t >= H : t < H) ? t : -1;
    };

    // rows of this worker (resident until the gradient: x, then 1 + exp(-x)); rounds beyond NG are neither
    // loaded nor used.  (Every load is consumed on every path that issued it: otherwise the compiler guards the
    // first later use with a vmcnt wait, which in P3 would wait for the gradient stores in flight.)
    float v[kPipeRounds][2][CH];
    if (w >= 2) {
#pragma unroll
        for (int g = 0; g < kPipeRounds; ++g) {
            if (g < NG) {                                    // (uniform)
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const int t = slot_row(g, side);
                    const float *row = p.x + (int64_t)(t >= 0 ? t : 0) * p.st + (int64_t)b * p.sb;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int c = lane + 64 * j;
                        v[g][side][j] = row[c < p.C ? c : p.C - 1];
                    }
                }
            }
        }
    }

    // y[b] -> LDS image [SP][PD], zero beyond S rows / C columns (MFMA K and N padding)
    const float *yb = p.y + (int64_t)b * p.S * p.C;
    for (int i = tid; i < p.SP * PD; i += kBinThreads) {
        const int l = i / PD, c = i - l * PD;
        sm.ys[i] = (l < p.S && c < p.C) ? yb[l * p.C + c] : 0.f;
    }
    if (tid < 32) sm.dummy[tid] = 0.f;                       // (also zeroes every flag)
    if (tid >= 64 && tid < 128) sm.zrow[tid - 64] = 0.f;
    for (int i = tid; i < kPrefetch * p.SP; i += kBinThreads) {
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }
    int64_t Tb64, L64;
    lens.get(Tb64, L64);
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;
    const float invC = 1.0f / (float)p.C;
    const float gs = p.grad_scale * invC;
    stamp(p, 1);

    // ---------------- the two scans ----------------
    if (w < 2) {
        __syncthreads();
        if (w == 1 && !p.grad && !GAMMA) return;
        __builtin_amdgcn_s_setprio(3);
        bool starved = false;
        if (Tb > 0) {
            int known = -1;                                  // steps whose emission rows are known to be in LDS
            auto ready = [&](int i) {
                while (known < i && !starved) {              // (uniform) the tile of step known + 1
                    const int t = w == 0 ? known + 1 : Tb - 2 - known;
                    const int m = t >> 4;
                    int spins = 0;
                    while (*(lds_cvint *)(sm.tile + m) == 0) {
                        if (++spins >= (1 << 22)) { starved = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    lds_order();
                    known = w == 0 ? 16 * m + 15 : Tb - 1 - 16 * m;   // the scan's last step inside this tile
                }
            };
            const bool rot = p.SP <= 63;
            if (w == 0) {
                if (rot) lattice_chain_fed<true, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, sm.prog, ready);
                else lattice_chain_fed<true, false>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, sm.prog, ready);
            } else {
                if (rot) lattice_chain_fed<false, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, sm.prog + 1, ready);
                else lattice_chain_fed<false, false>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, sm.prog + 1, ready);
            }
            if (starved) {                                   // let the workers through; they poison their rows
                *sm.fail = 1;
                lds_order();
                sm.prog[w] = Tb;
            }
        }
        __builtin_amdgcn_s_setprio(0);
        stamp(p, 5);
        if (w == 0) {
            float nll = ok ? -sm.al[(Tb - 1) * p.SP + (L - 1)] : -kNeg;
            if (starved) {                                   // a bounded wait ran out: NaN, not a plausible number
                nll = __builtin_nanf("");
                raise_status(p.counter, kStatusBinaryStarved);
            }
            publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        }
        stamp(p, 7);
        return;
    }

    // ---------------- workers, P1a: the logs of the rows -> D image, Q ----------------
    // With s = 1 + exp(-x):  p = 1/s,  log p = -log s,  log(1 - p) = -log s - x,  d = log p - log(1 - p) = x.
    // Usual case, every element of the row in (-16, 6) (no clamp can fire, 1 - p keeps its bits):
    //     Q = sum_c log(1 - p_c) = -ln2 * log2(prod_j s_j) - sum_j x_j  per lane (the product of <= 4 values
    //     below 9e6 cannot overflow), one exponential per element and one logarithm per lane; the reciprocal
    //     waits until the gradient.  log(1 - p) is exact here where the reference rounds 1 - fl(p) first (they
    //     differ by <= 2.4e-5 at x = 6: 1e-6 of an emission after the mean over C -- the exact (float64) value is the
    //     closer one); for x < 0 the cancellation in -log s - x costs <= 2e-6 absolute per element.
    // Otherwise: bce_logs_fast per element (the reference's clamps and its rounding of 1 - p).
    unsigned slow = 0;                                       // bit 2g+side: that row took the careful path
#pragma unroll
    for (int g = 0; g < kPipeRounds; ++g) {
        if (g < NG) {                                        // (uniform)
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int t = slot_row(g, side);
                float far = 0.f;                             // max_j |x_j + 5|: < 11 <=> every x_j in (-16, 6)
#pragma unroll
                for (int j = 0; j < CH; ++j) far = fmaxf(far, fabsf(v[g][side][j] + 5.0f));
                float ql;
                if (__builtin_amdgcn_ballot_w64(!(far < 11.0f)) != 0) slow |= 1u << (2 * g + side);
                if (!((slow >> (2 * g + side)) & 1)) {
                    float prod = 1.0f, sx = 0.f;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int c = lane + 64 * j;
                        const float xv = v[g][side][j];
                        const float sv = 1.0f + __builtin_amdgcn_exp2f(xv * -kLog2e);
                        v[g][side][j] = sv;
                        const bool in = j < CH - 1 || c < p.C;          // (only the last chunk can pass C)
                        prod *= in ? sv : 1.0f;
                        sx += in ? xv : 0.f;
                        if (t >= 0 && (j < CH - 1 || c < PD)) sm.dimg[t * PD + c] = in ? xv : 0.f;   // zero K padding
                    }
                    ql = __builtin_fmaf(__builtin_amdgcn_logf(prod), -kLn2, -sx);
                } else {
                    ql = 0.f;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int c = lane + 64 * j;
                        float sv, lp, lq;
                        bce_logs_fast_s(v[g][side][j], sv, lp, lq);
                        v[g][side][j] = sv;
                        const bool in = j < CH - 1 || c < p.C;
                        ql += in ? lq : 0.f;
                        if (t >= 0 && (j < CH - 1 || c < PD)) sm.dimg[t * PD + c] = in ? lp - lq : 0.f;
                    }
                }
                ql = wave_sum(ql);
                if (lane == 0 && t >= 0) sm.q[t] = ql;
            }
        }
    }
    stamp(p, 2);
    __syncthreads();
    stamp(p, 8);

    // ---------------- workers, P1b: emission tiles, the two ends of the sequence first ----------------
    bool starved = false;
    // this worker's place in the tile order (the SIMDs of workers 2, 3, 6, 7 carry two tile jobs at ten tiles, the others
    // three: the two end tiles, which the scans wait for, go to the lighter ones)
    const int i = u < 8 ? (u ^ 2) : u;                       // (28.1 -> 27.85 us at config 3)
    if (i < MT) {
        // tiles in the order the scans want them: 0, MT-1, 1, MT-2, ...; consecutive workers sit on different SIMDs.
        // (Tried and dropped: raised wave priority for the end tiles -- it does not order a SIMD's matrix pipe;
        // running the end tiles alone and the others behind them -- a lone job leaves the pipe idle during its
        // LDS waits and the middle tiles arrive late: 31.9 vs 27.9 us; prefetching the next K batch: 28.7 us.)
        const int m = (i & 1) ? MT - 1 - (i >> 1) : (i >> 1);
        const int NT = (p.SP + 15) >> 4, KT = (p.C + 3) >> 2;
        const int fr = lane & 15, fq = lane >> 4;
        const int ta = 16 * m + fr;
        const float *ap = sm.dimg + (ta < p.T ? ta : p.T - 1) * PD + fq;
        const float *bp[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int lrow = 16 * n + fr;
            bp[n] = sm.ys + (lrow < p.SP ? lrow : p.SP - 1) * PD + fq;
        }
        f32x4 acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto tile_job = [&](auto N) {                        // N column tiles share every A fragment
            constexpr int NN = decltype(N)::value;
            int kk = 0;
            for (; kk + 4 <= KT; kk += 4) {                  // fragment reads of four K steps in flight, then the MFMAs
                float fa[4], fb[4][NN];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = ap[4 * (kk + i)];
#pragma unroll
                    for (int n = 0; n < NN; ++n) fb[i][n] = bp[n][4 * (kk + i)];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int n = 0; n < NN; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[i][n], acc[n], 0, 0, 0);
            }
            for (; kk < KT; ++kk) {
                const float fa = ap[4 * kk];
#pragma unroll
                for (int n = 0; n < NN; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, bp[n][4 * kk], acc[n], 0, 0, 0);
            }
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                const int lrow = 16 * n + fr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 16 * m + 4 * fq + i;
                    if (t < Tb && lrow < p.SP) sm.em[t * p.SP + lrow] = lrow < L ? (acc[n][i] + sm.q[t]) * invC : kNeg;
                }
            }
        };
        using std::integral_constant;
        switch (NT) {
            case 1: tile_job(integral_constant<int, 1>{}); break;
            case 2: tile_job(integral_constant<int, 2>{}); break;
            case 3: tile_job(integral_constant<int, 3>{}); break;
            default: tile_job(integral_constant<int, 4>{}); break;
        }
        lds_order();
        sm.tile[m] = 1;                                      // (every lane, same value)
        __builtin_amdgcn_s_setprio(0);
    }
    stamp(p, 3);
    if (!p.grad && !GAMMA) return;

    // ---------------- workers, P3: gradient rows, four at a time ----------------
    //   gamma rows (wave-local LDS), then G = gamma . Y on v_mfma_f32_4x4x1 (16 blocks of 4x4, K = 1 per
    //   instruction): block = lane>>2 and column-in-block = lane&3 make the output column equal the lane and put
    //   the four rows in the four accumulator registers -- exactly the layout of the resident rows, so the
    //   gradient needs no LDS transpose.  A = gamma[row lane&3][l], B = Y[l][lane + 64 j].
    //   Every row's normaliser equals exp(-nll): only the first group takes the row maximum, the later ones
    //   shift by that group's log-normaliser (and still normalise by their own sum).  gamma carries the
    //   gradient scale, so an element costs a reciprocal (sigmoid), one fma and its store.
    //   This phase is bound by VALU issue (3.5 workers per SIMD): the code below is written for instruction count.
    bool have_lse = false;
    float c2 = 0.f;                                          // -lse * log2(e)
    const unsigned voff = 4u * lane;
    const int lane_l = lane < p.SP ? lane : 0;
    const bool in_l = lane < L;
    const float ninf = -__builtin_inff();
    // (one copy of the group code, run three times: the rows of the current group sit in v[4], v[5] and the others
    // rotate into place -- a third of the instruction footprint of three unrolled groups)
    auto group = [&](int jg) {
        if (2 * jg >= NG) return;                            // (uniform)
        int tt[4], tl[4];                                    // the rows of the group; the live ones among them
        int t_hi = -1, t_lo = p.T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tt[i] = slot_row(2 * jg + (i & 1), i >> 1);
            tl[i] = tt[i] < Tb ? tt[i] : -1;
            if (tl[i] >= 0) {
                t_hi = tl[i] > t_hi ? tl[i] : t_hi;
                t_lo = tl[i] < t_lo ? tl[i] : t_lo;
            }
        }
        f32x4 acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t_hi >= 0) {                                     // wave-uniform: at least one live row
            int spins = 0;
            for (;;) {                                       // (the three flags travel together: one LDS round trip per poll)
                const int pa = *(lds_cvint *)sm.prog, pb = *(lds_cvint *)(sm.prog + 1), bad = *(lds_cvint *)sm.fail;
                if (bad) starved = true;
                if (pa >= t_hi + 1 && pb >= Tb - t_lo) break;
                if (++spins >= (1 << 20)) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();
            if (jg == 2) stamp(p, 9);
            // posteriors of the four rows (one label per lane), gamma * scale over `be`
            // (idle slots read row 0 and are masked: twelve loads in flight, one wait)
            float z[4], pe[4], sum[4];
            {
                float za[4], zb[4], ze[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int off = (tl[k] >= 0 ? tl[k] : 0) * p.SP + lane_l;
                    za[k] = sm.al[off]; zb[k] = sm.be[off]; ze[k] = sm.em[off];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) z[k] = (in_l && tl[k] >= 0) ? za[k] + zb[k] - ze[k] : ninf;
            }
            if (!have_lse) {                                 // (uniform) first group of this worker
                float m[4] = {z[0], z[1], z[2], z[3]};
                wave_max4(m[0], m[1], m[2], m[3]);
#pragma unroll
                for (int k = 0; k < 4; ++k) pe[k] = __builtin_amdgcn_exp2f((z[k] - (tl[k] >= 0 ? m[k] : 0.f)) * kLog2e);
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[k] = pe[k];
                wave_sum4(sum[0], sum[1], sum[2], sum[3]);
#pragma unroll
                for (int k = 3; k >= 0; --k)
                    if (tl[k] >= 0) c2 = -kLog2e * m[k] - __builtin_amdgcn_logf(sum[k]);   // (uniform)
                have_lse = true;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) pe[k] = __builtin_amdgcn_exp2f(__builtin_fmaf(z[k], kLog2e, c2));
#pragma unroll
                for (int k = 0; k < 4; ++k) sum[k] = pe[k];
                wave_sum4(sum[0], sum[1], sum[2], sum[3]);
            }
            if (lane < p.SP) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tl[k] >= 0) sm.be[tl[k] * p.SP + lane] = pe[k] * (gs * __builtin_amdgcn_rcpf(sum[k]));
            }
            if (GAMMA && lane < p.S) {                       // posteriors output: gamma_t(l), rows sum to 1
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tl[k] >= 0)
                        p.gamma[((int64_t)b * p.T + tl[k]) * p.S + lane] = starved ? __builtin_nanf("") : pe[k] * __builtin_amdgcn_rcpf(sum[k]);
            }
            if (jg == 2) stamp(p, 10);
            const int ti = tl[lane & 3];
            const float *arow = ti >= 0 ? sm.be + ti * p.SP : sm.zrow;   // (an idle slot contributes a row of zeros)
            const float *yrow = sm.ys + lane;
            // four labels per batch, the next batch's operands in flight under this batch's MFMAs
            // (SP % 4 == 0; gamma and Y rows beyond L / S are 0; the last prefetch is clamped to rows that exist)
            float4 fa = *reinterpret_cast<const float4 *>(arow);
            float fy[4][CH];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CH; ++j) fy[i][j] = yrow[i * PD + 64 * j];
            for (int l0 = 0; l0 < (GAMMA ? 0 : L); l0 += 4) {
                const float4 fan = *reinterpret_cast<const float4 *>(arow + (l0 + 4 < p.SP ? l0 + 4 : 0));
                float fyn[4][CH];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < CH; ++j) fyn[i][j] = yrow[(l0 + 4 + i < p.SP ? l0 + 4 + i : p.SP - 1) * PD + 64 * j];
                const float fav[4] = {fa.x, fa.y, fa.z, fa.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(fav[i], fy[i][j], acc[j], 0, 0, 0);
                fa = fan;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < CH; ++j) fy[i][j] = fyn[i][j];
            }
            if (jg == 2) stamp(p, 11);
        }
        if (starved) raise_status(p.counter, kStatusBinaryStarved);
        if constexpr (GAMMA) {                               // posteriors only: rows beyond T_b get zeros, no gradient
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tt[i] >= 0 && tl[i] < 0 && lane < p.S) p.gamma[((int64_t)b * p.T + tt[i]) * p.S + lane] = 0.f;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tt[i] < 0) continue;                         // wave-uniform
            float *g = p.grad + ((int64_t)tt[i] * p.B + b) * p.C;
            bin_row_base_settled(g);
            const int slot = 2 * (2 * jg + (i & 1)) + (i >> 1);
            if (tl[i] < 0 || starved) {                      // (uniform) dead row: zeros; starved: NaN
                const float fill = starved ? __builtin_nanf("") : 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, fill);
            } else if ((slow >> slot) & 1) {                 // (uniform) a row with tails: torch's floored denominator
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float pr = __builtin_amdgcn_rcpf(v[4 + (i & 1)][i >> 1][j]);
                    const float pq = pr * (1.0f - pr);
                    const float gv = __builtin_fmaf(pr, gs, -acc[j][i]) * (pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f)));
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, gv);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float pr = __builtin_amdgcn_rcpf(v[4 + (i & 1)][i >> 1][j]);
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, __builtin_fmaf(pr, gs, -acc[j][i]));
                }
            }
        }
    };
    stamp(p, 6);
#pragma nounroll
    for (int jg = 2; jg >= 0; --jg) {                        // rows nearest the middle are ready first
        group(jg);
#pragma unroll
        for (int side = 0; side < 2; ++side)
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                v[4][side][j] = v[2][side][j]; v[5][side][j] = v[3][side][j];
                v[2][side][j] = v[0][side][j]; v[3][side][j] = v[1][side][j];
            }
    }
    stamp(p, 7);
}

template <bool WT>
static int launch_binary_pipe_wt(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int PD)
{
    const dim3 grid(p.B), block(kBinThreads);
    switch (ch) {
        case 1: return launch<binary_pipe_kernel<1, WT>>(grid, block, smem, s, p, PD);
        case 2: return launch<binary_pipe_kernel<2, WT>>(grid, block, smem, s, p, PD);
        case 3: return launch<binary_pipe_kernel<3, WT>>(grid, block, smem, s, p, PD);
        default: return launch<binary_pipe_kernel<4, WT>>(grid, block, smem, s, p, PD);
    }
}

static int launch_binary_pipe(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int PD)
{
    const bool wt = (size_t)8 * p.T * p.B * p.C <= ((size_t)230 << 20);   // logits + gradient within the memory-side cache
    return wt ? launch_binary_pipe_wt<true>(ch, smem, s, p, PD) : launch_binary_pipe_wt<false>(ch, smem, s, p, PD);
}

#include "binary_flow.hpp"

template <int K>
static int launch_binary_mfma(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int Tpad, int PD)
{
    const dim3 grid(p.B), block(kBinThreads);
    const bool wt = (size_t)8 * p.T * p.B * p.C <= ((size_t)230 << 20);   // logits + gradient within the memory-side cache
#define CTC_BIN_CASE(N)                                                                                        \
    case N: return wt ? launch<binary_mfma_kernel<K, N, true>>(grid, block, smem, s, p, Tpad, PD)               \
                      : launch<binary_mfma_kernel<K, N, false>>(grid, block, smem, s, p, Tpad, PD);
    switch (ch) {
        CTC_BIN_CASE(1) CTC_BIN_CASE(2) CTC_BIN_CASE(3)
        default: return wt ? launch<binary_mfma_kernel<K, 4, true>>(grid, block, smem, s, p, Tpad, PD)
                           : launch<binary_mfma_kernel<K, 4, false>>(grid, block, smem, s, p, Tpad, PD);
    }
#undef CTC_BIN_CASE
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
    p.nll = nll; p.loss = loss; p.grad = grad; p.gamma = nullptr;
    p.counter = static_cast<unsigned *>(workspace);
    static const int debug_stop = diag_env("CTC_AMD_DEBUG_STOP");
    static const bool binary_valu = diag_env("CTC_AMD_BINARY_VALU") != 0;
    p.stop = debug_stop < 0 ? debug_stop : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // fast path: MFMA contractions, rows resident (T <= 160, C <= 256, images fit in LDS)
    if (T <= (K == 1 ? kPipeMaxT : kBinRows * kBinWaves) && C <= 256 && !binary_valu) {
        BinaryParams q = p;
        q.SP = (p.SP + 3) / 4 * 4;                           // K padding of the gamma . Y product
        if (q.SP % K) q.SP = (q.SP + 4 * K - 1) / (4 * K) * (4 * K);
        const int Tpad = (T + 15) / 16 * 16;
        int PD = (C + 3) / 4 * 4 + 2;                        // >= C (+K padding), PD = 2 (mod 32):
        while (PD % 32 != 2) PD += 2;                        // conflict-free MFMA fragment reads
        const size_t need = binary_mfma_smem_bytes(T, Tpad, q.SP, PD);
        const int ch = (C + kWave - 1) / kWave;
        static const bool no_pipe = diag_env("CTC_AMD_BINARY_NOPIPE") != 0;
        static const bool no_flow = diag_env("CTC_AMD_BINARY_NOFLOW") != 0;
        if (K == 1 && !no_pipe && !no_flow && T <= kFlowMaxT && ch <= 3) {   // (four column chunks: 4 registers over the budget)   // the streamed kernel (binary_flow.hpp) has its own image pitch
            const int PF = binary_flow_pitch(C);
            const size_t fb = binary_flow_smem_bytes(T, q.SP, PF, C);
            if (fb <= kMaxLds) return launch_binary_flow(ch, fb, s, q, PF);
        }
        if (K == 1 && !no_pipe && binary_pipe_smem_bytes(T, q.SP, PD) <= kMaxLds)
            return launch_binary_pipe(ch, binary_pipe_smem_bytes(T, q.SP, PD), s, q, PD);
        if (need <= kMaxLds && T <= kBinRows * kBinWaves) {
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


// Per-step posteriors of the binary lattice: gamma[b,t,l] = P(label row l at step t | x, targets), the
// quantity the loss gradient contracts with the target rows -- it falls out of the pipelined kernel's gradient phase.
// Shapes of that kernel only (S <= 64, T <= 168, C <= 256, images within LDS); others: CTC_AMD_ERR_UNSUPPORTED_SHAPE.
extern "C" int ctc_amd_binary_posteriors(const float *x, int64_t stride_t, int64_t stride_b, const float *y,
                                         const int64_t *in_len, const int64_t *tgt_len, int T, int B, int C, int S,
                                         float *nll, float *gamma, void *workspace, void *stream)
{
    if (!x || !y || !in_len || !tgt_len || !nll || !gamma || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (S > kWave || T > kPipeMaxT || C > 256) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    BinaryParams q;
    q.x = x; q.st = stride_t; q.sb = stride_b; q.y = y;
    q.in_len = in_len; q.tgt_len = tgt_len;
    q.T = T; q.B = B; q.C = C; q.S = S;
    q.SP = (S + 3) / 4 * 4;
    q.CP = C | 1;
    q.loss_scale = 0.f; q.grad_scale = 0.f;
    q.nll = nll; q.grad = nullptr; q.gamma = gamma;
    q.counter = static_cast<unsigned *>(workspace);
    q.loss = reinterpret_cast<float *>(static_cast<char *>(workspace) + 32);   // the batch sum lands in a spare header word
    q.stop = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ch = (C + kWave - 1) / kWave;
    if (T <= kFlowMaxT && ch <= 3) {                          // the streamed kernel, cut off after the posteriors
        const int PF = binary_flow_pitch(C);
        const size_t fb = binary_flow_smem_bytes(T, q.SP, PF, C);
        if (fb <= kMaxLds) return launch_binary_flow_wt<true, true>(ch, fb, s, q, PF);
    }
    int PD = (C + 3) / 4 * 4 + 2;
    while (PD % 32 != 2) PD += 2;
    const size_t smem = binary_pipe_smem_bytes(T, q.SP, PD);
    if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    const dim3 grid(B), block(kBinThreads);
    switch (ch) {
        case 1: return launch<binary_pipe_kernel<1, true, true>>(grid, block, smem, s, q, PD);
        case 2: return launch<binary_pipe_kernel<2, true, true>>(grid, block, smem, s, q, PD);
        case 3: return launch<binary_pipe_kernel<3, true, true>>(grid, block, smem, s, q, PD);
        default: return launch<binary_pipe_kernel<4, true, true>>(grid, block, smem, s, q, PD);
    }
}
