// "K / M": the no-blank kernel of noblank_r16.hpp with each lattice chain split over TWO waves (included by
// noblank.hip behind noblank_r16.hpp, whose row type, DPP reductions and lattice idea it shares).
//
// Why.  At B = #CUs the launch is the serial lattice chain: 149 dependent steps, and a step
// of the (mantissa, exponent) recurrence is ten instruction slots of ONE wave (eight VALU + two LDS) -- 70 cycles,
// not because anything waits but because a wave issues one instruction every ~5-6 cycles.  The mantissa side of a
// step is two instructions; the rest is exponent bookkeeping.  Here the bookkeeping moves to a wave of its own:
//
//   K wave (one per direction)  a max-plus recurrence on FLOATS over the emissions' logarithms,
//         Kf_t(l) = max(Kf_{t-1}(l), Kf_{t-1}(l -+ 1)) + e_t(l),        e = log2 of the emission,
//     i.e. the log2 mass of the best path into the cell.  Its floor k_t(l) is the cell's exponent.  Nothing of the
//     mantissas feeds back into it: the cell value is alpha = m 2^k with m = alpha / 2^k, and alpha lies between the
//     best path's mass and (number of paths) times it, so m stays in [2^-eps, 2 N 2^eps] for ANY choice of k that
//     tracks the best path to within eps bits -- no renormalisation, no hand-back, the K wave runs ahead freely.
//     (N <= C(T-1, S-1); the host takes this kernel only when that fits fp32 with room for the product of two
//     mantissas, and starts the recurrence at +log2(N)/2 so that the mantissas are centred on 1.)
//   M wave (one per direction)  m_t(l) = m_{t-1}(l) pm 2^(k_{t-1}(l) - c) + m_{t-1}(l -+ 1) pm 2^(k_{t-1}(l -+ 1) - c),
//     c = k_t(l) - floor(e): two subtractions, two v_ldexp, a multiply and a DPP multiply-add per step, none of
//     which except the last two depends on the step before (the K wave stores (k, c) per cell).
//
// The arithmetic of a cell is the r16 kernel's (one multiply-add of two mantissas scaled by exact powers of two),
// only the exponent a cell is stored with differs.  Emissions below 2^-4096 of their row's maximum are clamped there
// (r16: 2^-(2^21)), which keeps the float recurrence exact to a few bits over 168 steps.
//
// 16 waves: K alpha, M alpha, K beta, M beta (one per SIMD) and 12 workers of four row groups each.  States sit on
// lanes 1..SP of a chain wave (lane 0 and the lanes behind them are idle: zero emission mantissa, a sinking
// exponent), so that no real state ever reads a DPP lane that does not exist.
#pragma once
#include "noblank_km_asm.hpp"

namespace ctc {

constexpr int kKmWorkers = 12, kKmG = 4;                    // 12 x 4 groups x 4 rows = 192 row slots
constexpr int kKmPad = 8;                                   // cells in front of t = 0 and behind t = T - 1 (even: pairs start at even t)
constexpr float kKmMinLog2 = -4096.f;                       // clamp of an emission's log2
constexpr float kKmSink = -8388608.f;                       // -2^23: "emission" of cells that carry no mass (idle lanes, masked states, pads)
constexpr float kKmNone = -16777216.f;                      // -2^24: Kf of a state no path has reached yet
constexpr int kKmIntMin = -2147483647 - 1;
constexpr int kKmKprog = kKmWorkers, kKmMprog = kKmWorkers + 2;   // sm.cnt: [0,12) workers' slots, [12,14) K progress, [14,16) M progress

__host__ __device__ inline int km_pitch_c(int T)            // 8-byte cells: = 2 (mod 4) -> 16-byte aligned rows, banks spread
{
    int tp = T + 2 * kKmPad;
    while ((tp & 3) != 2) ++tp;
    return tp;
}
__host__ __device__ inline int km_pitch_m(int T)            // 4-byte cells: = 4 (mod 8)
{
    int tp = T + 2 * kKmPad;
    while ((tp & 7) != 4) ++tp;
    return tp;
}

// row of slot k (0..3) of group g of worker u, or -1 (r16_row with 12 workers)
__device__ __forceinline__ int km_row(int T, int u, int g, int k)
{
    const int H = (T + 1) >> 1, idx = 2 * kKmWorkers * g + 2 * u + (k >> 1);
    if ((k & 1) == 0) return idx < H ? idx : -1;
    return idx < T - H ? T - 1 - idx : -1;
}

struct KmSmem {
    int TPc, TPm;
    cell_t *em;                                              // (pm, e): emission mantissa 2^frac(e), e = log2 emission (float)
    cell_t *ka, *kb;                                         // (k, c) as int bits: exponent of the cell, k - floor(e)
    float *ma, *mb;                                          // mantissas
    float *dummy, *stage, *cs;
    int *cnt, *lab, *occ, *done;
    __device__ KmSmem(float *base, int T, int SP, int RP)
    {
        TPc = km_pitch_c(T);
        TPm = km_pitch_m(T);
        cell_t *lat = reinterpret_cast<cell_t *>(base);
        const size_t A = (size_t)(SP + 1) * TPc;
        em = lat + kKmPad;
        ka = em + A;
        kb = ka + A;
        ma = reinterpret_cast<float *>(lat + 3 * A) + kKmPad;
        mb = ma + (size_t)(SP + 1) * TPm;
        dummy = mb + (size_t)(SP + 1) * TPm - kKmPad;        // 8 spare floats
        cnt = reinterpret_cast<int *>(dummy + 8);
        lab = cnt + 16;
        occ = lab + ((SP + 3) & ~3);
        cs = reinterpret_cast<float *>(occ + ((SP + 3) & ~3) + 4);
        done = reinterpret_cast<int *>(cs + 4 * kKmWorkers);
        stage = cs + 64;
    }
};

static size_t km_smem_bytes(int T, int SP, int C)
{
    const int RP = 32 * ((C + 31) / 32);
    return (size_t)3 * (SP + 1) * km_pitch_c(T) * 8 + (size_t)2 * (SP + 1) * km_pitch_m(T) * 4 +
           (8 + 16 + 2 * ((SP + 3) & ~3) + 4 + 64) * 4 + (size_t)kKmWorkers * 4 * RP * 4;
}

// log2 C(T-1, S-1) bounds the paths into any cell; the kernel is taken while two centred mantissas still multiply
// inside fp32 -- returns the offset the K recurrence starts with, or -1 for "do not take it"
static int km_offset(int T, int S)
{
    int n = T - 1, k = S - 1;
    if (k > n / 2) k = n / 2;
    const double lg = (lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0)) / 0.6931471805599453;
    if (lg > 88.0) return -1;
    return (int)(lg / 2.0 + 1.0);
}

__device__ __forceinline__ unsigned km_lds_addr(const void *p)   // byte offset in LDS of a pointer into the dynamic shared memory
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

__device__ __forceinline__ int km_flr(float x) { return (int)__builtin_floorf(x); }

template <bool FWD>
__device__ __forceinline__ float km_nbf(float v)             // neighbour state's value (lanes that have none read 0.0)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), FWD ? 0x138 : 0x130, 0xf, 0xf, true));
}
template <bool FWD>
__device__ __forceinline__ int km_nbi(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, FWD ? 0x138 : 0x130, 0xf, 0xf, true);
}

// what both chain waves share: which row a step works on, and the wait for the workers' groups
template <bool FWD>
struct KmWalk {
    int T, Tb, pos0, own, have;
    const int *cp;
    bool starved;
    __device__ KmWalk(const KmSmem &sm, int T_, int Tb_) : T(T_), Tb(Tb_), have(0), starved(false)
    {
        const int H = (T + 1) >> 1;
        pos0 = FWD ? 0 : T - Tb;
        own = FWD ? H : T - H;
        cp = sm.cnt + (lane_id() < kKmWorkers ? lane_id() : 0);
    }
    __device__ __forceinline__ int row(int i) const { return FWD ? i : Tb - 1 - i; }
    // the emission rows of steps 0..i_last must be published
    __device__ __forceinline__ void wait_upto(int i_last)
    {
        typedef const volatile __attribute__((address_space(3))) int lds_cvint;
        const int q = pos0 + (i_last < Tb ? i_last : Tb - 1);
        int ng = q / (2 * kKmWorkers) + 1;
        if (q >= own || ng > kKmG) ng = kKmG;
        if (ng <= have) return;
        const int need = lane_id() < kKmWorkers ? 4 * ng : 0;
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(*(lds_cvint *)cp < need) != 0) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_order();
        have = ng;
    }
};

typedef float km_pair_t __attribute__((ext_vector_type(4)));   // two 8-byte cells: (lower, higher) time index

// ---- K wave: exponents.  Stores (k, c) for every cell of its direction and publishes "steps done" in sm.cnt.
template <bool FWD>
__device__ __forceinline__ bool km_kchain(const NoblankParams &p, const KmSmem &sm, int T, int Tb, int L, int SP)
{
    const int lane = lane_id();
    const bool real = lane >= 1 && lane <= SP;
    const int lrow = real ? lane - 1 : SP;
    const cell_t *erow = sm.em + (size_t)lrow * sm.TPc;
    cell_t *orow = (FWD ? sm.ka : sm.kb) + (size_t)lrow * sm.TPc;
    int *prog = sm.cnt + kKmKprog + (FWD ? 0 : 1);
    KmWalk<FWD> wk(sm, T, Tb);
    const unsigned long long live = (4ull << SP) - 1;        // lanes 0 .. SP + 1: the states and a guard lane on either side
    __builtin_amdgcn_s_setprio(3);
    wk.wait_upto(4);
    float Kf;
    auto out = [&](float e) {                                // the cell of the step just taken
        const int k = km_flr(Kf);
        return make_cell(__builtin_bit_cast(float, k), k - km_flr(e));
    };
    auto step = [&](float e) { Kf = fmaxf(Kf, km_nbf<FWD>(Kf)) + e; };
    {                                                        // step 0: the start state alone
        const int x = wk.row(0);
        const float e0 = erow[x].y;
        Kf = lane == (FWD ? 1 : L) ? e0 + (float)p.koff : kKmNone;
        orow[x] = out(e0);
    }
    int i = 1;
    // single steps until the next two rows are a 16-byte pair (lower row even)
    if (i < Tb && ((FWD ? wk.row(i) : wk.row(i) - 1) & 1) != 0) {
        const int x = wk.row(i);
        const float e = erow[x].y;
        step(e);
        orow[x] = out(e);
        ++i;
    }
    lds_order();
    *prog = i;
    if (CTC_DIAG(p) < 0) stamp(p, 2);                        // diagnostic: the chain starts
    // double blocks of 16 steps through the hand-scheduled loop (tools/gen_km_asm.py), as many at once as the
    // workers have published rows for; what is left (< 16 steps) goes through the plain code below
    while (Tb - i >= 16) {
        int nb = (Tb - i) >> 4;
        if (wk.have < kKmG) {
            wk.wait_upto(i + 15);
            if (wk.have < kKmG) {                            // steps whose rows the published groups cover
                const int avail = 2 * kKmWorkers * wk.have - wk.pos0;
                nb = min(nb, max(1, (avail - i) >> 4));
            }
        }
        const int xlo = FWD ? wk.row(i) : wk.row(i) - 7;     // lowest row of the first block of eight steps
        unsigned ea = km_lds_addr(erow + xlo), oa = km_lds_addr(orow + xlo);
        const unsigned pa = km_lds_addr(prog);
        int pc = i, nbs = nb;
        if (FWD)
            asm volatile(CTC_KM_K_FWD : [ea] "+v"(ea), [oa] "+v"(oa), [kf] "+v"(Kf), [pc] "+v"(pc), [nb] "+s"(nbs) : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
        else
            asm volatile(CTC_KM_K_BWD : [ea] "+v"(ea), [oa] "+v"(oa), [kf] "+v"(Kf), [pc] "+v"(pc), [nb] "+s"(nbs) : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
        i += 16 * nb;
        if (CTC_DIAG(p) < 0) stamp(p, i < 112 ? 2 + (i >> 4) : 8);   // diagnostic: 16-step blocks done -> slots 3..8
    }
    constexpr int kBlk = 8;                                  // steps per look / publication
    while (i + 1 < Tb) {
        const int n = min(kBlk, (Tb - i) & ~1);              // an even number of steps
        if (wk.have < kKmG) wk.wait_upto(i + n - 1);
        const int xlo = FWD ? wk.row(i) : wk.row(i + n - 1); // lowest row of the block
        const km_pair_t *rb = reinterpret_cast<const km_pair_t *>(erow + xlo);
        km_pair_t *wb = reinterpret_cast<km_pair_t *>(orow + xlo);
        if (n == kBlk) {
            km_pair_t e[kBlk / 2];
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) e[q] = rb[FWD ? q : kBlk / 2 - 1 - q];
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) {
                const float e1 = FWD ? e[q].y : e[q].w, e2 = FWD ? e[q].w : e[q].y;
                step(e1);
                const cell_t c1 = out(e1);
                step(e2);
                const cell_t c2 = out(e2);
                km_pair_t o;
                o.x = FWD ? c1.x : c2.x;  o.y = FWD ? c1.y : c2.y;
                o.z = FWD ? c2.x : c1.x;  o.w = FWD ? c2.y : c1.y;
                wb[FWD ? q : kBlk / 2 - 1 - q] = o;
            }
        } else {
            for (int q = 0; q < n / 2; ++q) {
                const km_pair_t ee = rb[FWD ? q : n / 2 - 1 - q];
                const float e1 = FWD ? ee.y : ee.w, e2 = FWD ? ee.w : ee.y;
                step(e1);
                const cell_t c1 = out(e1);
                step(e2);
                const cell_t c2 = out(e2);
                km_pair_t o;
                o.x = FWD ? c1.x : c2.x;  o.y = FWD ? c1.y : c2.y;
                o.z = FWD ? c2.x : c1.x;  o.w = FWD ? c2.y : c1.y;
                wb[FWD ? q : n / 2 - 1 - q] = o;
            }
        }
        i += n;
        lds_order();
        *prog = i;
    }
    if (i < Tb) {                                            // one row left
        if (wk.have < kKmG) wk.wait_upto(i);
        const int x = wk.row(i);
        const float e = erow[x].y;
        step(e);
        orow[x] = out(e);
        ++i;
        lds_order();
        *prog = i;
    }
    __builtin_amdgcn_s_setprio(0);
    return wk.starved;
}

// ---- M wave: mantissas, behind its K wave.  Returns the cell alpha[T_b-1, L_b-1] (FWD) as (mantissa, exponent).
template <bool FWD>
__device__ __forceinline__ cell_t km_mchain(const NoblankParams &p, const KmSmem &sm, int T, int Tb, int L, int SP)
{
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    const int lane = lane_id();
    const bool real = lane >= 1 && lane <= SP;
    const int lrow = real ? lane - 1 : SP;
    const cell_t *erow = sm.em + (size_t)lrow * sm.TPc;
    const cell_t *krow = (FWD ? sm.ka : sm.kb) + (size_t)lrow * sm.TPc;
    float *orow = (FWD ? sm.ma : sm.mb) + (size_t)lrow * sm.TPm;
    const int *kprog = sm.cnt + kKmKprog + (FWD ? 0 : 1);
    int *prog = sm.cnt + kKmMprog + (FWD ? 0 : 1);
    const KmWalk<FWD> wk(sm, T, Tb);
    const unsigned long long live = (4ull << SP) - 1;
    bool starved = false;
    int seen = 0;                                            // steps the K wave is known to have finished
    auto wait_k = [&](int steps) {
        if (steps <= seen) return;
        int spins = 0;
        while ((seen = *(lds_cvint *)kprog) < steps) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_order();
    };
    __builtin_amdgcn_s_setprio(3);
    float m;
    int kprev;
    auto step = [&](float pm, int k, int c) {
        const int d1 = kprev - c, d2 = km_nbi<FWD>(kprev) - c;
        const float s1 = __builtin_amdgcn_ldexpf(pm, d1), s2 = __builtin_amdgcn_ldexpf(pm, d2);
        m = __builtin_fmaf(km_nbf<FWD>(m), s2, m * s1);
        kprev = k;
    };
    wait_k(1);
#ifdef CTC_AMD_FAULT_INJECT                                  // tests/test_status.py: sample 0's alpha chain "starves"
    if (FWD && blockIdx.x == 0) starved = true;
#endif
    {                                                        // step 0: alpha = p on the start state, as m 2^k
        const int x = wk.row(0);
        const cell_t kc = krow[x];
        const float pm = erow[x].x;
        m = lane == (FWD ? 1 : L) ? __builtin_amdgcn_ldexpf(pm, -cell_k(kc)) : 0.f;
        kprev = __builtin_bit_cast(int, kc.x);
        orow[x] = m;
    }
    int i = 1;
    if (i < Tb && ((FWD ? wk.row(i) : wk.row(i) - 1) & 1) != 0) {
        wait_k(i + 1);
        const int x = wk.row(i);
        const cell_t kc = krow[x];
        step(erow[x].x, __builtin_bit_cast(int, kc.x), cell_k(kc));
        orow[x] = m;
        ++i;
    }
    lds_order();
    *prog = i;
    if (CTC_DIAG(p) < 0) stamp(p, 2);
    // double blocks of 16 steps through the hand-scheduled loop, as many at once as the K wave has finished
    while (Tb - i >= 16) {
        int nb = (min(seen, Tb) - i) >> 4;
        if (nb < 1) {
            wait_k(i + 16);
            if (starved) break;
            continue;
        }
        const int xlo = FWD ? wk.row(i) : wk.row(i) - 7;
        unsigned ka = km_lds_addr(krow + xlo), ea = km_lds_addr(erow + xlo), oa = km_lds_addr(orow + xlo);
        const unsigned pa = km_lds_addr(prog);
        int pc = i, nbs = nb;
        if (FWD)
            asm volatile(CTC_KM_M_FWD : [ka] "+v"(ka), [ea] "+v"(ea), [oa] "+v"(oa), [m] "+v"(m), [kp] "+v"(kprev), [pc] "+v"(pc), [nb] "+s"(nbs)
                         : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
        else
            asm volatile(CTC_KM_M_BWD : [ka] "+v"(ka), [ea] "+v"(ea), [oa] "+v"(oa), [m] "+v"(m), [kp] "+v"(kprev), [pc] "+v"(pc), [nb] "+s"(nbs)
                         : [pa] "v"(pa), [mk] "s"(live) : CTC_KM_CLOBBERS);
        i += 16 * nb;
        if (CTC_DIAG(p) < 0) stamp(p, i < 112 ? 2 + (i >> 4) : 8);
    }
    constexpr int kBlk = 8;
    while (i + 1 < Tb) {
        const int n = min(kBlk, (Tb - i) & ~1);
        wait_k(i + n);
        const int xlo = FWD ? wk.row(i) : wk.row(i + n - 1);
        const km_pair_t *kb_ = reinterpret_cast<const km_pair_t *>(krow + xlo);
        const km_pair_t *eb_ = reinterpret_cast<const km_pair_t *>(erow + xlo);
        f2_t *wb = reinterpret_cast<f2_t *>(orow + xlo);
        if (n == kBlk) {
            km_pair_t kc[kBlk / 2], ee[kBlk / 2];
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) {
                kc[q] = kb_[FWD ? q : kBlk / 2 - 1 - q];
                ee[q] = eb_[FWD ? q : kBlk / 2 - 1 - q];
            }
#pragma unroll
            for (int q = 0; q < kBlk / 2; ++q) {
                const float pm1 = FWD ? ee[q].x : ee[q].z, pm2 = FWD ? ee[q].z : ee[q].x;
                const int k1 = __builtin_bit_cast(int, FWD ? kc[q].x : kc[q].z), c1 = __builtin_bit_cast(int, FWD ? kc[q].y : kc[q].w);
                const int k2 = __builtin_bit_cast(int, FWD ? kc[q].z : kc[q].x), c2 = __builtin_bit_cast(int, FWD ? kc[q].w : kc[q].y);
                step(pm1, k1, c1);
                const float m1 = m;
                step(pm2, k2, c2);
                f2_t o;
                o.x = FWD ? m1 : m;
                o.y = FWD ? m : m1;
                wb[FWD ? q : kBlk / 2 - 1 - q] = o;
            }
        } else {
            for (int q = 0; q < n / 2; ++q) {
                const km_pair_t kq = kb_[FWD ? q : n / 2 - 1 - q], eq = eb_[FWD ? q : n / 2 - 1 - q];
                const float pm1 = FWD ? eq.x : eq.z, pm2 = FWD ? eq.z : eq.x;
                const int k1 = __builtin_bit_cast(int, FWD ? kq.x : kq.z), c1 = __builtin_bit_cast(int, FWD ? kq.y : kq.w);
                const int k2 = __builtin_bit_cast(int, FWD ? kq.z : kq.x), c2 = __builtin_bit_cast(int, FWD ? kq.w : kq.y);
                step(pm1, k1, c1);
                const float m1 = m;
                step(pm2, k2, c2);
                f2_t o;
                o.x = FWD ? m1 : m;
                o.y = FWD ? m : m1;
                wb[FWD ? q : n / 2 - 1 - q] = o;
            }
        }
        i += n;
        lds_order();
        *prog = i;
    }
    if (i < Tb) {
        wait_k(i + 1);
        const int x = wk.row(i);
        const cell_t kc = krow[x];
        step(erow[x].x, __builtin_bit_cast(int, kc.x), cell_k(kc));
        orow[x] = m;
        ++i;
        lds_order();
        *prog = i;
    }
    __builtin_amdgcn_s_setprio(0);
    // alpha[T_b-1, L_b-1] from where the chains stored it
    const int fx = FWD ? Tb - 1 : 0;
    const float fm = (FWD ? sm.ma : sm.mb)[(size_t)(L - 1) * sm.TPm + fx];
    const cell_t fkc = (FWD ? sm.ka : sm.kb)[(size_t)(L - 1) * sm.TPc + fx];
    cell_t fin = make_cell(fm, __builtin_bit_cast(int, fkc.x));
    if (starved) {
        raise_status(p.counter, kStatusNoblankStarved);
        fin = make_cell(__builtin_nanf(""), 0);
    }
    return fin;
}

template <int N4, int N2, bool NT>
__global__ __launch_bounds__(kThreads, 4) void noblank_km_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    typedef R16Row<N4, N2> Row;
    constexpr int RP = Row::kCols;
    constexpr int G = kKmG;
    const KmSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, RP);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 0) note_arrival(p.counter, b);                // (a K wave has no other vector-memory operation)
    const int u = w - 4;                                     // waves 0..3: K alpha, M alpha, K beta, M beta; 4..15: workers 0..11
    const int rho = lane >> 4, i16 = lane & 15;
    const float ninf = -__builtin_inff();

    if (CTC_DIAG(p) == 1) return;
    stamp(p, 0);
    const ScalarLengths len(p.in_len + b, p.tgt_len + b);
    int tv[G];
    Row v[G];
    const bool col_ok = Row::off_last(i16) < p.C;
    const int c_last = col_ok ? Row::off_last(i16) : p.C - (Row::kLast4 ? 4 : 2);
    if (u >= 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tv[g] = km_row(p.T, u, g, rho);
            v[g].load(row_ptr(p, tv[g] >= 0 ? tv[g] : 0, b), i16, c_last);
        }
    }
    int64_t Tb64, L64;
    len.get(Tb64, L64);
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;
    if (w == 0) {                                            // labels: one wave's business, in a branch of its own
        int k = 0;
        if (lane < L) {
            k = load_label(p.lab, p.lab64, (int64_t)b * p.S + lane) % p.C;
            if (k < 0) k += p.C;                             // python negative index (NoBlankCTC.py:102)
        }
        if (lane < p.SP) sm.lab[lane] = k;
    }
    // emission cells that nobody publishes (pads, spare row, rows beyond the sequence) carry no mass
    const cell_t none = make_cell(0.f, 0);
    {
        cell_t nn = none;
        nn.y = kKmSink;
        for (int i = tid; i < (p.SP + 1) * sm.TPc; i += kThreads) sm.em[i - kKmPad] = nn;
        // the spare row of the exponent cells is what the idle lanes of the M waves read before anything is written there
        const cell_t kn = make_cell(__builtin_bit_cast(float, -(1 << 23)), 0);
        for (int i = tid; i < sm.TPc; i += kThreads) {
            sm.ka[(size_t)p.SP * sm.TPc + i - kKmPad] = kn;
            sm.kb[(size_t)p.SP * sm.TPc + i - kKmPad] = kn;
        }
    }
    if (tid < 16) sm.cnt[tid] = 0;
    if (tid == 16) *sm.done = 0;
    if (tid < 8) sm.dummy[tid] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(p, 1);
    if (CTC_DIAG(p) == 2) return;

    if (Tb == 0) {                                           // no alignment exists: nll = 1e13, zero gradient
        if (w == 1) publish_and_reduce_sum(-kNeg, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        if (u >= 0 && p.grad) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (tv[g] >= 0) Row::template store_zero<NT>(p.grad + ((int64_t)tv[g] * p.B + b) * p.C, i16, col_ok);
        }
        return;
    }
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;

    // ---------------------------------------------------------------- chain waves
    if (u < 0) {
        if (w == 0) {
            if (km_kchain<true>(p, sm, p.T, Tb, L, p.SP)) raise_status(p.counter, kStatusNoblankStarved);
            stamp(p, 9);
        } else if (w == 2) {
            if (p.grad && km_kchain<false>(p, sm, p.T, Tb, L, p.SP)) raise_status(p.counter, kStatusNoblankStarved);
            stamp(p, 9);
        } else if (w == 1) {
            const cell_t a = km_mchain<true>(p, sm, p.T, Tb, L, p.SP);
            stamp(p, 11);
            // nll = -log alpha[T_b-1, L_b-1] (NoBlankCTC.py:58-68,139) plus the per-row constants the chains left out
            bool late = false;
            {
                int spins = 0;
                while (*(lds_cvint *)sm.done < kKmWorkers) {
                    if (++spins >= kSpinLimit) { late = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                lds_order();
            }
            const float csum = wave_sum(lane < 4 * kKmWorkers ? sm.cs[lane] : 0.f);
            const float am = a.x;
            float nll = am > 0.f ? -(__builtin_amdgcn_logf(am) + (float)cell_k(a) + csum) * kLn2 : -kNeg;
            if (am != am) nll = am;
            if (late) {
                raise_status(p.counter, kStatusNoblankStarved);
                nll = __builtin_nanf("");
            }
            publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        } else if (p.grad) {
            {   // M beta has to wait for its K wave anyway: the occurrence index of every state among equal labels
                // (0 = first; the workers add repeated labels up in one pass per repetition, P3)
                const int kl = lane < p.SP ? sm.lab[lane] : -1;
                int oc = 0;
                for (int l2 = 0; l2 < L; ++l2) {
                    const int q = __builtin_amdgcn_readlane(kl, l2);
                    oc += (l2 < lane && q == kl) ? 1 : 0;
                }
                if (lane >= L) oc = 0;
                int mo = 0;
                while (__builtin_amdgcn_ballot_w64(oc > mo) != 0) ++mo;
                if (lane < p.SP) sm.occ[lane] = oc;
                sm.occ[(p.SP + 3) & ~3] = mo;
                lds_order();
            }
            km_mchain<false>(p, sm, p.T, Tb, L, p.SP);
            stamp(p, 11);
        }
        return;
    }

    // ---------------------------------------------------------------- workers
    float *tile = sm.stage + (size_t)u * 4 * RP;
    float *trow = tile + rho * RP;
    int lst[2];
    float *gat[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        lst[s] = 16 * s + i16;
        gat[s] = trow + (lst[s] < p.SP ? sm.lab[lst[s]] : 0);
    }
    const bool own[2] = {lst[0] < L, lst[1] < L};
    const bool smooth = p.ls_b != 0.f;
    cell_t *const spare_w = reinterpret_cast<cell_t *>(sm.dummy);
    const int Hh = (p.T + 1) >> 1;
    bool grp[G];
#pragma unroll
    for (int g = 0; g < G; ++g) grp[g] = 2 * kKmWorkers * g + 2 * u < Hh;
    float mrow[G];
    // P1a: the emissions the chains wait for (see noblank_r16.hpp): e = (x[lab_l] - max) log2e (times a when smoothed)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
        if (grp[g]) {
            Row &x = v[g];
            const int t = tv[g];
            const bool live = t >= 0 && t < Tb;
            float m = x.max();
            row16_allmax(m);
            mrow[g] = m;
            if (CTC_DIAG(p) < 0 && g == 0) stamp(p, 6);
            x.to_tile(trow, i16);
            lds_order();
            float xv[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) xv[s] = *gat[s];
            lds_order();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float ec = (xv[s] - m) * kLog2e;
                const float e2 = fmaxf(smooth ? p.ls_a * ec : ec, kKmMinLog2);
                const float pm = __builtin_amdgcn_exp2f(e2 - __builtin_floorf(e2));
                cell_t *dst = (live && lst[s] < p.SP) ? sm.em + lst[s] * sm.TPc + t : spare_w;
                cell_t cc;
                cc.x = own[s] ? pm : 0.f;
                cc.y = own[s] ? e2 : kKmSink;
                *dst = cc;
            }
            lds_order();
        } else {
            mrow[g] = 0.f;
        }
        sm.cnt[u] = 4 * (g + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (CTC_DIAG(p) < 0 && g < 3) stamp(p, 8 + g);
    }
    __builtin_amdgcn_s_setprio(0);
    // P1b: exp(x - max), row sums, the per-row constants of the loss
    float rs[G];
    float cacc = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rs[g] = 0.f;
        if (!grp[g]) continue;
        Row &x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tb;
        const float m = mrow[g];
        float sx = 0.f;
        if (smooth) {
            sx = x.sum(col_ok);
            row16_allsum(sx);
        }
        const float mb = -m * kLog2e;
        float sum = x.exp_sum(mb, col_ok ? mb : ninf);
        row16_allsum(sum);
        const float l2sum = __builtin_amdgcn_logf(sum);
        rs[g] = live ? p.grad_scale * __builtin_amdgcn_rcpf(sum) : 0.f;
        const float cst = smooth ? __builtin_fmaf(p.ls_b, __builtin_fmaf(sx - (float)p.C * m, kLog2e, -(float)p.C * l2sum), -p.ls_a * l2sum)
                                 : -l2sum;
        cacc += live ? cst : 0.f;
    }
    if (i16 == 0) sm.cs[4 * u + rho] = cacc;
    lds_order();
    if (lane == 0) __hip_atomic_fetch_add(sm.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    stamp(p, 2);
    float prefetched = 0.f;
    if (p.next_round > 0 && (int)blockIdx.x + p.next_round < p.B) {
        const int nb = xcd_sample(blockIdx.x + p.next_round, p.B);
        const int line = u * kWave + lane;
        const int lines_per_row = (p.C * 4 + 127) / 128;
        const int t = line / lines_per_row, c = (line - t * lines_per_row) * 32;
        typedef const float __attribute__((address_space(1))) gfloat;
        if (t < p.T) prefetched = *(gfloat *)(row_ptr(p, t, nb) + (c < p.C ? c : p.C - 1));
    }
    if (!p.grad) return;

    const int Tlive = Tb;
    const float gsc = p.grad_scale;
    bool starved = false;

    // P3: middle-out (see noblank_r16.hpp); gamma ~ alpha q / p with alpha = ma 2^ka, q = mb 2^kb, 1 / p = 2^-floor(e) / pm
    int kref = 0;
    bool have_ref = false;
    bool first = true;
#pragma unroll
    for (int g = G - 1; g >= 0; --g) {
        if (!grp[g]) continue;
        int need_a = 0, need_b = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tk = km_row(p.T, u, g, k);
            if (tk >= 0 && tk < Tlive) {
                need_a = max(need_a, tk + 1);
                need_b = max(need_b, Tlive - tk);
            }
        }
        if (need_a > 0) {
            int spins = 0;
            while (*(lds_cvint *)(sm.cnt + kKmMprog) < need_a || *(lds_cvint *)(sm.cnt + kKmMprog + 1) < need_b) {
                if (++spins >= kSpinLimit) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();
        }
#ifdef CTC_AMD_FAULT_INJECT
        if (b == 1) starved = true;
#endif
        if (CTC_DIAG(p) < 0) stamp(p, 3 + (G - 1 - g) < 6 ? 3 + (G - 1 - g) : 5);
        if (first) { asm volatile("" ::"v"(prefetched)); first = false; }
        int occn[2] = {0, 0}, max_occ = 0;
        if (need_a > 0) {
            occn[0] = own[0] ? sm.occ[lst[0]] : 0;
            occn[1] = own[1] ? sm.occ[lst[1]] : 0;
            max_occ = __builtin_amdgcn_readfirstlane(sm.occ[(p.SP + 3) & ~3]);
        }
        const Row &x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tlive;
        float pr[2];
        int ks[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool in = live && own[s];
            const int offc = in ? lst[s] * sm.TPc + t : p.SP * sm.TPc, offm = in ? lst[s] * sm.TPm + t : p.SP * sm.TPm;
            const float a = sm.ma[offm], bq = sm.mb[offm];
            const int ca = cell_k(sm.ka[offc]);               // k_a - floor(e)
            const int kbq = __builtin_bit_cast(int, sm.kb[offc].x);
            const float pm = in ? sm.em[offc].x : 1.f;
            pr[s] = in ? a * bq * __builtin_amdgcn_rcpf(pm) : 0.f;
            ks[s] = ca + kbq;
        }
        if (__builtin_amdgcn_ballot_w64(live && !have_ref) != 0) {
            int km = max(pr[0] > 0.f ? ks[0] : kKmIntMin, pr[1] > 0.f ? ks[1] : kKmIntMin);
            row16_allmax(km);
            if (live && !have_ref && km != kKmIntMin) { kref = km; have_ref = true; }
        }
        float z[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) z[s] = pr[s] > 0.f ? __builtin_amdgcn_ldexpf(pr[s], ks[s] - kref) : 0.f;
        float tot = z[0] + z[1];
        row16_allsum(tot);
        float rinv = (live && tot > 0.f) ? -gsc * p.ls_a * __builtin_amdgcn_rcpf(tot) : 0.f;   // (store_grad adds the tile: MINUS the occupancy)
        if (starved) {
            rinv = __builtin_nanf("");
            raise_status(p.counter, kStatusNoblankStarved);
        }
        Row::zero_tile(trow, i16);
        lds_order();
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (own[s] && occn[s] == 0) *gat[s] = z[s] * rinv;
        for (int k = 1; k <= max_occ; ++k) {
            lds_order();
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (own[s] && occn[s] == k) *gat[s] += z[s] * rinv;
        }
        lds_order();
        if (t >= 0) {
            float *gp = p.grad + ((int64_t)t * p.B + b) * p.C;
            if (smooth) x.template store_grad<NT, true>(gp, trow, i16, rs[g] * (1.f - p.ls_b), col_ok, live ? -p.ls_b * gsc : 0.f);
            else x.template store_grad<NT, false>(gp, trow, i16, rs[g], col_ok, 0.f);
        }
        lds_order();
    }
    stamp(p, 7);
}

}  // namespace ctc
