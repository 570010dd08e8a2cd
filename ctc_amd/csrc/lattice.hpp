// The T x S no-blank lattice scans (alpha, beta') shared by the no-blank and binary
// kernels.  One wave per chain, K consecutive label states per lane (S <= 64*K),
// neighbour state through one DPP wave shift, emissions prefetched from LDS.
//
// Recursion (computes_transition, NoBlankCTC.py:71-87):
//   alpha_t(l) = LSE(alpha_{t-1}(l), alpha_{t-1}(l-1)) + e_t(l), advance dropped on the
//   first step (:75-76), cells l >= L_b forced to the sentinel (:79-80).
// beta'_t(l) = beta_t(l) + e_t(l) is its time/label mirror started at (T_b-1, L_b-1)
// (the reference's own beta pass is dead code, :113-125; autograd does that work).
#pragma once
#include "common.hpp"

namespace ctc {

constexpr int kPrefetch = 4;   // emission rows in flight ahead of the chain

// em, out: [T][SP] in LDS.  FWD: t = 0..Tb-1 from state 0; !FWD: t = Tb-1..0 from L-1.
template <int K, bool FWD>
__device__ __forceinline__ void lattice_chain(const float *em, float *out, int Tb, int L, int SP)
{
    const int l0 = lane_id() * K;
    const int start = FWD ? 0 : L - 1;
    float a[K];
    float ring[kPrefetch][K];

    auto row_of = [&](int i) { return FWD ? i : Tb - 1 - i; };

    {   // first step: only "stay" from the virtual start state
        const int t = row_of(0);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int l = l0 + k;
            a[k] = (l == start) ? em[t * SP + l] : kNeg;
            if (l < SP) out[t * SP + l] = a[k];
        }
    }
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) {
        const int i = 1 + j;
#pragma unroll
        for (int k = 0; k < K; ++k)
            ring[j][k] = (i < Tb && l0 + k < SP) ? em[row_of(i) * SP + l0 + k] : 0.f;
    }
    for (int i0 = 1; i0 < Tb; i0 += kPrefetch) {
#pragma unroll
        for (int j = 0; j < kPrefetch; ++j) {
            const int i = i0 + j;
            if (i < Tb) {                                   // wave-uniform
                const int t = row_of(i);
                float e[K];
#pragma unroll
                for (int k = 0; k < K; ++k) e[k] = ring[j][k];
                const int in = i + kPrefetch;
#pragma unroll
                for (int k = 0; k < K; ++k)
                    ring[j][k] = (in < Tb && l0 + k < SP) ? em[row_of(in) * SP + l0 + k] : 0.f;
                const float nb = FWD ? wave_shr1(a[K - 1], kNeg) : wave_shl1(a[0], kNeg);
                float n[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float adv = FWD ? (k == 0 ? nb : a[k - 1]) : (k == K - 1 ? nb : a[k + 1]);
                    const float v = lse2(a[k], adv) + e[k];
                    n[k] = (l0 + k < L) ? v : kNeg;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    a[k] = n[k];
                    if (l0 + k < SP) out[t * SP + l0 + k] = a[k];
                }
            }
        }
    }
}

}  // namespace ctc
