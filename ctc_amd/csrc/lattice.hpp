// The T x S no-blank lattice scans (alpha, beta') shared by the no-blank and binary
// kernels.  One wave per chain, K consecutive label states per lane (S <= 64*K),
// neighbour state through one DPP wave shift, emissions prefetched from LDS.
//
// Recursion (computes_transition, NoBlankCTC.py:71-87):
//   alpha_t(l) = LSE(alpha_{t-1}(l), alpha_{t-1}(l-1)) + e_t(l), advance dropped on the
//   first step (:75-76), cells l >= L_b forced to the sentinel (:79-80).
// beta'_t(l) = beta_t(l) + e_t(l) is its time/label mirror started at (T_b-1, L_b-1)
// (the reference's own beta pass is dead code, :113-125; autograd does that work).
//
// LDS layout contract (LatticeSmem): em has kPrefetch pad rows on both sides so the
// prefetch may run past either end; emissions of masked cells (l >= L_b) are stored as
// the sentinel, which keeps those states "very negative" without a per-step select.
#pragma once
#include "common.hpp"

namespace ctc {

constexpr int kPrefetch = 4;   // emission rows in flight ahead of the chain

// em, out: [T][SP] in LDS (SP multiple of K), `dummy`: one spare float for idle lanes.
// FWD: t = 0..Tb-1 from state 0; !FWD: t = Tb-1..0 from state L-1.
// ROT: lane 63 is idle (SP <= 63*K), so a wave rotate needs no fill value.
// Serial dependency per step: dpp -> sub -> mul -> exp -> add -> log -> fma -> add.
// `prog` (optional, LDS): the number of completed steps is published there every kPrefetch
// steps, so that other waves can start on rows both scans have passed (binary.hip).
template <int K, bool FWD, bool ROT>
__device__ __forceinline__ void lattice_chain(const float *em, float *out, float *dummy, int Tb, int L, int SP,
                                              int *prog = nullptr)
{
    const int l0 = lane_id() * K;
    const bool act = l0 < SP;                       // whole lane inside or outside the row
    const int dir = FWD ? SP : -SP;
    const int t_first = FWD ? 0 : Tb - 1;
    // idle lanes (beyond the row) read the sentinel from a pad row and write a spare slot,
    // both with stride 0: their state stays "very negative" and can be rotated into lane 0
    const float *rd = act ? em + t_first * SP + l0 : em - kPrefetch * SP;
    float *wr = act ? out + t_first * SP + l0 : dummy;
    const int winc = act ? dir : 0;
    const int start = FWD ? 0 : L - 1;
    float a[K];
    float ring[kPrefetch][K];

    auto shift = [&]() {
        if (ROT) {
            const int v = __builtin_bit_cast(int, FWD ? a[K - 1] : a[0]);
            return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(v, FWD ? 0x13C : 0x134, 0xf, 0xf, false));
        }
        return FWD ? wave_shr1(a[K - 1], kNeg) : wave_shl1(a[0], kNeg);
    };
    auto step = [&](const float (&e)[K]) {
        const float nb = shift();
        float n[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float adv = FWD ? (k == 0 ? nb : a[k - 1]) : (k == K - 1 ? nb : a[k + 1]);
            const float t = __builtin_amdgcn_exp2f(-fabsf(a[k] - adv) * kLog2e);
            n[k] = __builtin_fmaf(__builtin_amdgcn_logf(1.0f + t), kLn2, vmax(a[k], adv)) + e[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { a[k] = n[k]; wr[k] = n[k]; }
        wr += winc;
    };

    // first step: only "stay" from the virtual start state
#pragma unroll
    for (int k = 0; k < K; ++k) {
        a[k] = (l0 + k == start) ? rd[k] : kNeg;
        wr[k] = a[k];
    }
    wr += winc;
    rd += winc;
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) {                   // rows 1..kPrefetch (pad rows absorb overrun)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = rd[k];
        rd += winc;
    }
    int i = 1;
    for (; i + kPrefetch <= Tb; i += kPrefetch) {           // branch-free body
        if (prog) { lds_order(); *prog = i; }               // (wave-uniform; every lane, same value)
#pragma unroll
        for (int j = 0; j < kPrefetch; ++j) {
            float e[K];
#pragma unroll
            for (int k = 0; k < K; ++k) { e[k] = ring[j][k]; ring[j][k] = rd[k]; }
            rd += winc;
            step(e);
        }
    }
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j)                      // tail: ring[j] holds row i+j
        if (i + j < Tb) step(ring[j]);
    if (prog) { lds_order(); *prog = Tb; }
}

// Posterior row: gamma_t(l) = exp(alpha_t(l) + beta'_t(l) - e_t(l)) / sum_l' (same), written
// over `be`.  Mathematically every row's normaliser equals exp(-nll); normalising per
// row cancels the rounding error the two long fp32 scans share (11x smaller error at
// T=150) and decouples the gradient from the final nll.  A row is handled by a group of
// G = 16/32/64 lanes; `t` may differ between the groups of a wave, `live` masks a group.
// Label positions that repeat a class are then folded into the first occurrence
// (dup[l] != 0 marks a first occurrence that has repeats, nxt[] chains them), so the
// gradient pass reads ONE value per class.  Wave-local: LDS ops of a wave are in order.
template <bool FOLD>
__device__ __forceinline__ void posterior_row(const float *al, float *be, const float *em, const int *nxt,
                                              const int *dup, int t, bool live, int L, int SP, int G)
{
    const int ll = lane_id() % G;
    const int off = t * SP;
    if (SP <= kWave) {                                   // one label per lane: stay in registers
        const bool in = live && ll < L;
        const float v = in ? al[off + ll] + be[off + ll] - em[off + ll] : -__builtin_inff();
        const float m = group_reduce<true>(v, G);
        const float pexp = in ? fast_exp(v - m) : 0.f;
        const float s = group_reduce<false>(pexp, G);
        if (live && ll < SP) be[off + ll] = pexp * (1.0f / s);
    } else {
        float m = -__builtin_inff();
        if (live)
            for (int l = ll; l < L; l += G) m = fmaxf(m, al[off + l] + be[off + l] - em[off + l]);
        m = group_reduce<true>(m, G);
        float s = 0.f;
        if (live)
            for (int l = ll; l < L; l += G) s += fast_exp(al[off + l] + be[off + l] - em[off + l] - m);
        s = group_reduce<false>(s, G);
        const float inv = 1.0f / s;
        if (live)
            for (int l = ll; l < SP; l += G)
                be[off + l] = (l < L) ? fast_exp(al[off + l] + be[off + l] - em[off + l] - m) * inv : 0.f;
    }
    if (FOLD && live)
        for (int l = ll; l < L; l += G)
            if (dup[l]) {
                float tot = be[off + l];
                for (int n = nxt[l]; n >= 0; n = nxt[n]) tot += be[off + n];
                be[off + l] = tot;
            }
}

// Four posterior rows at once for S <= 64 (one label per lane): the two reductions of each
// row are interleaved four-wide (wave_max4 / wave_sum4, common.hpp), gamma * `scale` is
// written over `be`.  t[k] < 0 marks an idle slot.  No repeat folding (binary variant).
__device__ __forceinline__ void posterior_rows4(const float *al, float *be, const float *em, const int (&t)[4],
                                                int L, int SP, float scale)
{
    const int lane = lane_id(), lcl = lane < SP ? lane : 0;
    const float ninf = -__builtin_inff();
    float z[4], pe[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int off = (t[k] >= 0 ? t[k] : 0) * SP + lcl;
        const float zz = al[off] + be[off] - em[off];
        z[k] = (lane < L && t[k] >= 0) ? zz : ninf;
        pe[k] = z[k];
    }
    wave_max4(z[0], z[1], z[2], z[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        pe[k] = __builtin_amdgcn_exp2f((pe[k] - (t[k] >= 0 ? z[k] : 0.f)) * kLog2e);
        z[k] = pe[k];
    }
    wave_sum4(z[0], z[1], z[2], z[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t[k] >= 0 && lane < SP) be[t[k] * SP + lane] = pe[k] * (scale * __builtin_amdgcn_rcpf(z[k]));
}

__device__ __forceinline__ int posterior_group(int SP) { return SP <= 16 ? 16 : (SP <= 32 ? 32 : 64); }

}  // namespace ctc
