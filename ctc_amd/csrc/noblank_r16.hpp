// "Sixteen-lane rows": the pipelined no-blank kernel with a lean worker (included by noblank.hip).
//
// Chains and hand-off are those of noblank_xr.hpp (extended-range linear lattice).  What changes is
// how the 14 worker waves hold their rows of x.  A single wave issues at most one instruction
// every ~4.4 cycles whatever its kind (tools/micro/issue_rate.hip), so a worker's time is its
// INSTRUCTION COUNT; the one-row-per-wave layout of noblank_pipe.hpp / noblank_xr.hpp spends
// ~250 instructions per row (two six-step whole-wave reductions, scalar bookkeeping per row,
// class-table lookups, folding of repeated labels).  Here a wave works on FOUR rows at a time,
// one row per 16-lane DPP row, each lane holding 2*CH2 elements as float2 pairs:
//
//   * loads / stores are 8-byte (global_load/store_dwordx2): 128 contiguous bytes per row per
//     instruction, CH2 instructions per four rows instead of 3 per row;
//   * row max / sum are serial over the lane's own elements plus a FOUR-step DPP all-reduce that
//     serves the four rows at once and leaves the result in every lane (no readlane, no scalar
//     per-row state: t, 1/sum, liveness are per-lane vectors);
//   * the emission gather x[t, lab_l] goes through a small LDS staging tile (the four raw rows
//     are written once, the 4 x S gathered values come back with two ds_read_b32);
//   * in the gradient pass the same tile becomes the class-occupancy tile: it is zeroed, the
//     scaled posteriors are scattered into it (first occurrences of a class store, the k-th
//     repetition of a label is added in a k-th plain read-modify-write pass -- LDS float atomics
//     held the LDS pipeline long enough to slow the chains), and read back 8 bytes per lane next
//     to the resident exp(x - max) values.
//
// About 45 instructions per row instead of ~250.  Needs C even (8-byte aligned rows), C <= 256,
// S <= 31, T <= 168 and 35 KB more LDS than noblank_xr.hpp (one workgroup per CU).
//
// Measured and NOT kept (git history): a first pass in plain doubles under one common
// scale per chain (4 instead of 11 VALU instructions per step) with a row-total self-check and this
// lattice as the fallback -- the chain is bound by its two LDS operations and its per-block
// bookkeeping, not by its arithmetic (89 against 100 cycles per step), and the check's barrier and
// the later loss ticket cost more than the chain gained (19.2 against 17.4 us at config 2).
#pragma once

namespace ctc {

typedef float f2_t __attribute__((ext_vector_type(2)));

#define CTC_ROW16(OP)                                                                   \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"       \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"            \
    "s_nop 1"
// all-reduce inside each 16-lane DPP row, result in every lane: four VALU instructions
__device__ __forceinline__ void row16_allmax(float &v) { asm volatile(CTC_ROW16("v_max_f32_dpp") : "+v"(v)); }
__device__ __forceinline__ void row16_allsum(float &v) { asm volatile(CTC_ROW16("v_add_f32_dpp") : "+v"(v)); }
__device__ __forceinline__ void row16_allmax(int &v) { asm volatile(CTC_ROW16("v_max_i32_dpp") : "+v"(v)); }

// Lattice layout of this kernel: TRANSPOSED, [state][time] with time contiguous -- cell t of state l
// sits at l * TP + kR16Pad + t (8-byte cells) in each of the three arrays.  A chain lane then walks
// its own row with a compile-time stride, so the loads / stores of an unrolled block of steps use
// immediate offsets (no address arithmetic per step), and TWO consecutive steps move as ONE 16-byte
// LDS access: the pad is odd so that the pairs (rows 1-2, 3-4, ...) start on 16-byte boundaries
// (the backward chain peels one single step off when T_b is odd, so that its pairs are the same).
// Pitch TP = 2 (mod 4) cells: 16-byte aligned rows, and the lanes of a chain access (4 banks each,
// stride 2 TP words) fall on different banks.  kR16Pad cells before t = 0 and after t = T-1 of every
// state row take the chains' prefetch (4 steps ahead) and the last, partly idle group of four steps
// (up to 3 steps past the end); `em` pads are zero.  One spare state row SP: all zeros in `em` (read
// by the chain lanes that only watch progress counters), scratch in `al` / `be`.
constexpr int kR16Pad = 2 * kPrefetch + 1;                  // odd
#ifndef CTC_R16_BLOCK
#define CTC_R16_BLOCK 16
#endif
constexpr int kR16Block = CTC_R16_BLOCK;                    // chain steps per block (one look at the workers, one renormalisation)
__host__ __device__ inline int r16_pitch(int T)
{
    int tp = T + 2 * kR16Pad + 1;
    while ((tp & 3) != 2) ++tp;
    return tp;
}

// Row of slot k (0..3) of group g of worker u, or -1: positions 28 g + 2 u and 28 g + 2 u + 1 of the
// front half (k = 0, 2: rows counted from t = 0) and of the back half (k = 1, 3: from t = T-1).  Group g
// of all workers together covers positions [28 g, 28 g + 28) of both halves, which is all the chains
// need to know; inside a group the workers' rows sit at DIFFERENT distances from the ends, so that in
// the gradient pass they become ready one worker after the other while the chains run out (worker 13
// 26 steps before the end, worker 0 at the end) instead of all fourteen at the very end.
__device__ __forceinline__ int r16_row(int T, int u, int g, int k)
{
    const int H = (T + 1) >> 1, idx = 2 * kPipeWorkers * g + 2 * u + (k >> 1);
    if ((k & 1) == 0) return idx < H ? idx : -1;
    return idx < T - H ? T - 1 - idx : -1;
}

struct R16Smem {
    int TP;
    cell_t *em, *al, *be;                                    // (mantissa, exponent) cells, [state][time]
    float *dummy, *stage, *cs;
    int *cnt, *lab, *occ, *done;
    __device__ R16Smem(float *base, int T, int SP, int RP)
    {
        TP = r16_pitch(T);
        cell_t *lat = reinterpret_cast<cell_t *>(base);
        em = lat + kR16Pad;                                  // -> cell (t = 0, l = 0)
        al = em + (size_t)(SP + 1) * TP;
        be = al + (size_t)(SP + 1) * TP;
        dummy = reinterpret_cast<float *>(lat + (size_t)3 * (SP + 1) * TP);   // write-only spare cells
        cnt = reinterpret_cast<int *>(dummy + 8);
        lab = cnt + 16;
        occ = lab + ((SP + 3) & ~3);                        // [SP] occurrence index, [SPpad] their maximum
        cs = reinterpret_cast<float *>(occ + ((SP + 3) & ~3) + 4);      // [workers][4 rows]: sum over the row slots of the per-row constants
        done = reinterpret_cast<int *>(cs + 4 * kPipeWorkers);          // workers whose constants are in `cs`
        stage = cs + 64;                                                 // [workers][4 rows][RP]
    }
};

static size_t r16_smem_bytes(int T, int SP, int C)
{
    const int RP = 32 * ((C + 31) / 32);
    return (size_t)3 * (SP + 1) * r16_pitch(T) * 8 + (8 + 16 + 2 * ((SP + 3) & ~3) + 4 + 64) * 4 +
           (size_t)kPipeWorkers * 4 * RP * 4;
}

// alpha (FWD) / beta (!FWD) chain of this kernel.  Both directions run the SAME recurrence on
// (mantissa, exponent) cells,
//     u_t(l) = p_t(l) * (u_prev(l) + u_prev(l -+ 1)),        u = alpha resp. q = beta * p,
// i.e. beta is carried WITH the emission of its own step (gamma_t(l) ~ alpha_t(l) q_t(l) / p_t(l), the
// division is the workers' business).  A step is split into an exponent side that never looks at a
// mantissa,
//     kk = max(k, k');  s = p_m 2^(k - kk);  s' = p_m 2^(k' - kk);  k = kk + p_k      (6 VALU)
// and a mantissa side of two dependent instructions, m = m s + m' s' (multiply + DPP multiply-add):
// the exponents form a max-plus recurrence of their own that runs ahead of the mantissas inside the
// wave, so the serial path of a step is two instructions instead of four and the other six fill
// its issue slots.  The larger term is scaled by p_m in [1, 2), so a mantissa never shrinks and grows
// at most fourfold per step: one v_frexp renormalisation per block of 16 steps keeps it in range (it
// is the only place where an exponent depends on a mantissa).
// Hand-off: a worker publishes its rows in groups of four slots; group g of ALL workers together
// covers positions [28 g, 28 g + 28) of each half of the sequence, so "rows up to position q are
// there" is one scalar: the number of groups every worker must have finished.  It changes three
// times per chain; a block of steps whose requirement is already met costs no look at all.
// Returns alpha[T_b-1, L_b-1] (FWD) as a cell.
template <bool FWD>
__device__ __forceinline__ cell_t r16_chain(const NoblankParams &p, const R16Smem &sm, int T, int Tb, int L, int SP, const int lane)
{
    constexpr int G = kPipeRows / 4;
    constexpr int D = FWD ? 1 : -1;
    // Lanes beyond the states stay ALIVE on the spare state row (zero emissions: no mass ever, and
    // nothing a real state reads): the same instruction stream runs faster with the whole wave
    // switched on than with 20 lanes (tools/micro/chain_asm.hip).  A backward lane reads its upper
    // neighbour, so state SP-1 sees the zero exponent of an idle lane; a forward lane reads its
    // lower neighbour, so idle lanes only ever copy exponents nobody looks at.
    const int lrow = lane < SP ? lane : SP;
    // step i = 0..T_b-1 works on row x(i) = FWD ? i : T_b - 1 - i: reads its emissions, writes its cells
    const cell_t *erow = sm.em + (size_t)lrow * sm.TP;
    cell_t *orow = (FWD ? sm.al : sm.be) + (size_t)lrow * sm.TP;
    typedef float pair_t __attribute__((ext_vector_type(4)));           // two cells: (lower, higher) time index
    float m;
    int k;

    const int H = (T + 1) >> 1;
    const int pos0 = FWD ? 0 : T - Tb;                       // position (in this chain's half order) of step 0
    const int own = FWD ? H : T - H;
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    const int *cp = sm.cnt + (lane < kPipeWorkers ? lane : 0);
    int have = 0;                                            // groups known to be published by every worker
    bool starved = false;
    auto wait_upto = [&](int i_last) {                       // rows of steps 0..i_last must be published
        const int q = pos0 + (i_last < Tb ? i_last : Tb - 1);
        int ng = q / (2 * kPipeWorkers) + 1;
        if (q >= own || ng > G) ng = G;
        if (ng <= have) return;                              // (scalar) nothing new to wait for
        const int need = lane < kPipeWorkers ? 4 * ng : 0;
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(*(lds_cvint *)cp < need) != 0) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        lds_order();
        have = ng;
    };
    // One time step with emission (em, ek), hand-ordered: eight VALU instructions, none of which needs a
    // wait state in front of it -- the new exponent is formed right behind the max, so that the next
    // step's DPP read of it (and this step's DPP read of the old one) are far enough from their
    // writers.  An `s_nop` in front of every v_max_i32_dpp (what the compiler makes of the plain C++
    // form, and what it puts between two asm statements) costs a full issue slot and more
    // (tools/micro/chain_asm.hip: 43 cycles per step with, 37 without), so a PAIR of steps is one asm
    // statement.  (`first`: exponent / mantissa were just written by compiler-scheduled code, which
    // knows nothing of the DPP reads in here: two wait states of our own.)
#define CTC_R16_STEP_ASM(K, M, EM, EK, K2, T2, DPP)                                                  \
            "v_max_i32_dpp %[kk], " K ", " K " " DPP "\n\t"                                           \
            "v_add_u32 " K2 ", %[kk], " EK "\n\t"                                                     \
            "v_sub_u32 %[d1], " K ", %[kk]\n\t"                                                       \
            "v_sub_u32_dpp %[d2], " K ", %[kk] " DPP "\n\t"                                           \
            "v_ldexp_f32 %[d1], " EM ", %[d1]\n\t"                                                    \
            "v_ldexp_f32 %[d2], " EM ", %[d2]\n\t"                                                    \
            "v_mul_f32 " T2 ", " M ", %[d1]\n\t"                                                      \
            "v_fmac_f32_dpp " T2 ", " M ", %[d2] " DPP "\n\t"
#ifdef CTC_X_ROWDPP                                           // experiment: what the cross-row shift costs (wrong beyond 16 states)
#define CTC_R16_DPP_F "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define CTC_R16_DPP_B "row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#else
#define CTC_R16_DPP_F "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define CTC_R16_DPP_B "wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
#endif
    auto step = [&](float em_, float ek_) {                  // a single step (alignment peel)
        int kk, k2, d1, d2;
        float t;
        if (FWD)
            asm("s_nop 1\n\t" CTC_R16_STEP_ASM("%[k]", "%[m]", "%[em]", "%[ek]", "%[k2]", "%[t]", CTC_R16_DPP_F)
                : [kk] "=&v"(kk), [k2] "=&v"(k2), [d1] "=&v"(d1), [d2] "=&v"(d2), [t] "=&v"(t)
                : [k] "v"(k), [m] "v"(m), [em] "v"(em_), [ek] "v"(ek_));
        else
            asm("s_nop 1\n\t" CTC_R16_STEP_ASM("%[k]", "%[m]", "%[em]", "%[ek]", "%[k2]", "%[t]", CTC_R16_DPP_B)
                : [kk] "=&v"(kk), [k2] "=&v"(k2), [d1] "=&v"(d1), [d2] "=&v"(d2), [t] "=&v"(t)
                : [k] "v"(k), [m] "v"(m), [em] "v"(em_), [ek] "v"(ek_));
        m = t;
        k = k2;
    };
    // steps (i, i+1) with the pair of emission cells `e` (lower, higher index); returns the pair to store
    auto step2 = [&](pair_t e, bool first) {
        const float e1m = FWD ? e.x : e.z, e1k = FWD ? e.y : e.w, e2m = FWD ? e.z : e.x, e2k = FWD ? e.w : e.y;
        int kk, d1, d2, k1, k2;
        float m1, m2;
#define CTC_R16_PAIR(NOP, DPP)                                                                       \
        asm(NOP CTC_R16_STEP_ASM("%[k]", "%[m]", "%[e1m]", "%[e1k]", "%[k1]", "%[m1]", DPP)           \
                CTC_R16_STEP_ASM("%[k1]", "%[m1]", "%[e2m]", "%[e2k]", "%[k2]", "%[m2]", DPP)         \
            : [kk] "=&v"(kk), [d1] "=&v"(d1), [d2] "=&v"(d2), [k1] "=&v"(k1), [k2] "=&v"(k2),        \
              [m1] "=&v"(m1), [m2] "=&v"(m2)                                                         \
            : [k] "v"(k), [m] "v"(m), [e1m] "v"(e1m), [e1k] "v"(e1k), [e2m] "v"(e2m), [e2k] "v"(e2k))
        if (FWD) { if (first) { CTC_R16_PAIR("s_nop 1\n\t", CTC_R16_DPP_F); } else { CTC_R16_PAIR("", CTC_R16_DPP_F); } }
        else { if (first) { CTC_R16_PAIR("s_nop 1\n\t", CTC_R16_DPP_B); } else { CTC_R16_PAIR("", CTC_R16_DPP_B); } }
#undef CTC_R16_PAIR
        m = m2;
        k = k2;
        const float k1f = __builtin_bit_cast(float, k1), k2f = __builtin_bit_cast(float, k2);
        pair_t o;
        o.x = FWD ? m1 : m2;  o.y = FWD ? k1f : k2f;         // lower index: the earlier step when walking up
        o.z = FWD ? m2 : m1;  o.w = FWD ? k2f : k1f;
        return o;
    };
    auto renorm = [&]() {                                    // mantissa back into [0.5, 1)  (0 stays 0)
        k += __builtin_amdgcn_frexp_expf(m);
        m = __builtin_amdgcn_frexp_mantf(m);
    };

    int *prog = sm.cnt + kPipeWorkers + (FWD ? 0 : 1);
#ifdef CTC_X_NOCHAIN
    *prog = Tb;                                              // experiment: what the launch costs without the chains
    return make_cell(1.f, kXrBias);
#endif
#ifndef CTC_X_NOPRIO
    __builtin_amdgcn_s_setprio(3);                           // the chains are the critical path
#endif
    wait_upto(kPrefetch + 1);
#ifdef CTC_AMD_FAULT_INJECT                                  // tests/test_status.py: sample 0's alpha chain "starves"
    if (FWD && blockIdx.x == 0) starved = true;
#endif
    int i = 1;
    {                                                        // step 0: u = p on the start state only
        const int start = FWD ? 0 : L - 1, x = FWD ? 0 : Tb - 1;
        const cell_t e0 = erow[x];
        m = lane == start ? e0.x : 0.f;
        k = lane == start ? kXrBias + cell_k(e0) : 0;
        orow[x] = make_cell(m, k);
        if (!FWD && (Tb & 1) == 1 && Tb > 1) {               // align the pairs: one single step (wave-uniform)
            const cell_t e = erow[Tb - 2];
            step(e.x, e.y);
            orow[Tb - 2] = make_cell(m, k);
            i = 2;
        }
    }
    // row of the current step: the lowest (FWD) / highest (!FWD) cell index of the pair (i, i+1)
    const int x0 = FWD ? i : Tb - 1 - i;
    constexpr int kPairs = kR16Block / 2;
    // block-relative bases at the LOWEST address a block touches, so that the unrolled pairs use
    // non-negative immediate offsets in both directions (ds offsets are unsigned).  The reads run
    // kPrefetch steps (two pairs) ahead of the writes.
    const pair_t *rb = reinterpret_cast<const pair_t *>(erow + (FWD ? x0 + kPrefetch : x0 - kPrefetch - (kR16Block - 1)));
    pair_t *wb = reinterpret_cast<pair_t *>(orow + (FWD ? x0 : x0 - (kR16Block - 1)));
    pair_t ring[2];
    {
        const pair_t *r0 = reinterpret_cast<const pair_t *>(erow + (FWD ? x0 : x0 - 3));
        ring[0] = r0[FWD ? 0 : 1];                           // steps i, i+1
        ring[1] = r0[FWD ? 1 : 0];                           // steps i+2, i+3
    }
    const unsigned long long loop_t0 = CTC_DIAG(p) == -77 ? __builtin_amdgcn_s_memtime() : 0;   // (chain probe only)
    for (; i + kR16Block <= Tb; i += kR16Block) {
        lds_order();
#ifndef CTC_X_NOPROG
        *prog = i;                                           // steps < i are done (every lane, same value)
#endif
        if (have < G) wait_upto(i + kR16Block - 1 + kPrefetch);
        if (CTC_DIAG(p) < 0) stamp(p, 2 + i / 16);       // diagnostic: block starts -> slots 2..10
#ifndef CTC_X_NORENORM
        renorm();
#endif
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
            const pair_t e = ring[q & 1];
#ifndef CTC_X_NOREAD
            ring[q & 1] = rb[FWD ? q : kPairs - 1 - q];
#endif
#ifdef CTC_X_NOWRITE
            const pair_t o = step2(e, q == 0);
            if (q == kPairs - 1) wb[0] = o;
#elif defined(CTC_X_WMASK)
            const pair_t o = step2(e, q == 0);
            if (lane < SP) wb[FWD ? q : kPairs - 1 - q] = o;  // experiment: idle lanes switched off for the store
#else
            wb[FWD ? q : kPairs - 1 - q] = step2(e, q == 0);
#endif
        }
        rb += D * kPairs;
        wb += D * kPairs;
    }
    if (CTC_DIAG(p) == -77 && lane == 0)
        reinterpret_cast<unsigned long long *>(p.counter)[FWD ? 0 : 1] = __builtin_amdgcn_s_memtime() - loop_t0;
    lds_order();
    *prog = i;
    if (have < G) wait_upto(Tb - 1);
    renorm();
    // the rest in groups of kPrefetch steps = one revolution of the ring (no guards inside).  The
    // last group may run up to three steps past the end: those read zero pad cells and write pad
    // cells of the output rows, which nobody looks at -- the final state is read back below.
    static_assert(kPrefetch == 4, "tail groups assume one ring revolution of two pairs");
    if (!FWD) { rb += kPairs - 2; wb += kPairs - 2; }        // lowest address of a 4-step group
    for (; i < Tb; i += kPrefetch) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const pair_t e = ring[q];
            ring[q] = rb[FWD ? q : 1 - q];
            wb[FWD ? q : 1 - q] = step2(e, q == 0);
        }
        rb += D * 2;
        wb += D * 2;
    }
    lds_order();
    *prog = Tb;
    __builtin_amdgcn_s_setprio(0);
    // alpha[T_b-1, L_b-1] from where the chain stored it (the registers may hold overrun steps);
    // a hand-off that ran out of patience poisons the sample instead of returning a plausible number
    cell_t fin = (FWD ? sm.al : sm.be)[(size_t)(L - 1) * sm.TP + (FWD ? Tb - 1 : 0)];
    if (starved) {
        raise_status(p.counter, kStatusNoblankStarved);
        fin = make_cell(__builtin_nanf(""), 0);
    }
    return fin;
}

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(8)));   // rows in global memory: 8-byte aligned only

// One lane's share of one row: N4 float4 (columns 64 j + 4 i .. + 3) and then N2 float2 (columns
// 64 N4 + 32 k + 2 i, + 1), i = position of the lane in its 16-lane DPP row.  16-byte accesses
// wherever the row allows: a vector memory instruction costs its issue slot whatever it carries
// (tools/micro/rw_phase.hip: the launch's whole read-then-write traffic takes 7.6 us in 8-byte and
// 5.8 us in 16-byte pieces), and the same goes for the LDS tile.  Only the LAST chunk can stick out
// of the row (C even; a float4 last chunk only when C % 4 == 0, see r16_shape).
template <int N4, int N2>
struct R16Row {
    f4_t a[N4 > 0 ? N4 : 1];
    f2_t c[N2 > 0 ? N2 : 1];
    static constexpr int kCols = 64 * N4 + 32 * N2;
    static constexpr bool kLast4 = N2 == 0;                  // the last chunk is a float4
    __device__ static __forceinline__ int off4(int j, int i16) { return 64 * j + 4 * i16; }
    __device__ static __forceinline__ int off2(int k, int i16) { return 64 * N4 + 32 * k + 2 * i16; }
    __device__ static __forceinline__ int off_last(int i16) { return kLast4 ? off4(N4 - 1, i16) : off2(N2 - 1, i16); }

    __device__ __forceinline__ void load(const float *row, int i16, int c_last)
    {
#pragma unroll
        for (int j = 0; j < N4; ++j)
            a[j] = *reinterpret_cast<const f4u_t *>(row + ((kLast4 && j == N4 - 1) ? c_last : off4(j, i16)));
#pragma unroll
        for (int k = 0; k < N2; ++k)
            c[k] = *reinterpret_cast<const f2_t *>(row + (k == N2 - 1 ? c_last : off2(k, i16)));
    }
    // largest element.  No mask: a lane whose last chunk would stick out of the row was given the row's LAST chunk
    // instead (`c_last`), and duplicates do not change a maximum.  Plain v_max3 / v_max (the values are loaded
    // logits: fmaxf would quiet every one of them first, a v_max x, x, x per element).
    __device__ __forceinline__ float max() const
    {
        constexpr int n = 4 * N4 + 2 * N2;
        float e[n];
#pragma unroll
        for (int j = 0; j < N4; ++j) { e[4 * j] = a[j].x; e[4 * j + 1] = a[j].y; e[4 * j + 2] = a[j].z; e[4 * j + 3] = a[j].w; }
#pragma unroll
        for (int k = 0; k < N2; ++k) { e[4 * N4 + 2 * k] = c[k].x; e[4 * N4 + 2 * k + 1] = c[k].y; }
        float m = e[0];
#pragma unroll
        for (int i = 1; i + 1 < n; i += 2) m = vmax3(m, e[i], e[i + 1]);
        if ((n & 1) == 0) m = vmax(m, e[n - 1]);
        return m;
    }
    // the lane's sum of the raw elements (`valid_last`: the last chunk lies inside the row)
    __device__ __forceinline__ float sum(bool valid_last) const
    {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const float q = (a[j].x + a[j].y) + (a[j].z + a[j].w);
            s += ((kLast4 && j == N4 - 1) && !valid_last) ? 0.f : q;
        }
#pragma unroll
        for (int k = 0; k < N2; ++k) s += (k == N2 - 1 && !valid_last) ? 0.f : c[k].x + c[k].y;
        return s;
    }
    // x <- exp2(x log2e + mb), the last chunk with `mbl` instead (= mb, or -inf for a lane whose last chunk is a
    // duplicate: exp2(-inf) = 0); returns the lane's sum.  The multiply-adds and the sums go through the packed
    // fp32 pipe (two elements per instruction), the exponentials are what they are (quarter rate).
    __device__ __forceinline__ float exp_sum(float mb, float mbl)
    {
        const f2_t l2 = {kLog2e, kLog2e};
        f2_t acc = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const float b = (kLast4 && j == N4 - 1) ? mbl : mb;
            const f2_t bb = {b, b};
            f2_t lo = {a[j].x, a[j].y}, hi = {a[j].z, a[j].w};
            lo = __builtin_elementwise_fma(lo, l2, bb);
            hi = __builtin_elementwise_fma(hi, l2, bb);
            a[j].x = __builtin_amdgcn_exp2f(lo.x);
            a[j].y = __builtin_amdgcn_exp2f(lo.y);
            a[j].z = __builtin_amdgcn_exp2f(hi.x);
            a[j].w = __builtin_amdgcn_exp2f(hi.y);
            const f2_t e0 = {a[j].x, a[j].y}, e1 = {a[j].z, a[j].w};
            acc += e0;
            acc += e1;
        }
#pragma unroll
        for (int k = 0; k < N2; ++k) {
            const float b = k == N2 - 1 ? mbl : mb;
            const f2_t bb = {b, b};
            const f2_t lo = __builtin_elementwise_fma(c[k], l2, bb);
            c[k].x = __builtin_amdgcn_exp2f(lo.x);
            c[k].y = __builtin_amdgcn_exp2f(lo.y);
            acc += c[k];
        }
        return acc.x + acc.y;
    }
    // the lane's columns of a staged row in LDS (`trow` = start of the row in the worker's tile)
    __device__ __forceinline__ void to_tile(float *trow, int i16) const
    {
#pragma unroll
        for (int j = 0; j < N4; ++j) *reinterpret_cast<f4_t *>(trow + off4(j, i16)) = a[j];
#pragma unroll
        for (int k = 0; k < N2; ++k) *reinterpret_cast<f2_t *>(trow + off2(k, i16)) = c[k];
    }
    __device__ static __forceinline__ void zero_tile(float *trow, int i16)
    {
        const f4_t z4 = {0.f, 0.f, 0.f, 0.f};
        const f2_t z2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < N4; ++j) *reinterpret_cast<f4_t *>(trow + off4(j, i16)) = z4;
#pragma unroll
        for (int k = 0; k < N2; ++k) *reinterpret_cast<f2_t *>(trow + off2(k, i16)) = z2;
    }
    // grad row = x * rs + tile, the tile holding MINUS the class occupancy (nothing to negate here), written through;
    // `g` = start of the row in grad (`cb`: a constant added to every element -- the label-smoothing term, 0 otherwise)
    template <bool NT, bool LS>
    __device__ __forceinline__ void store_grad(float *g, const float *trow, int i16, float rs, bool col_ok, float cb) const
    {
        const f2_t r2 = {rs, rs}, cb2 = {cb, cb};
        // every tile read first, then the arithmetic and the stores: a gradient store is inline asm with a memory clobber,
        // and a tile read placed behind one waits for its own LDS round trip with nothing else in flight -- three round
        // trips in a row per group
        f4_t o4[N4 > 0 ? N4 : 1];
        f2_t o2[N2 > 0 ? N2 : 1];
#pragma unroll
        for (int j = 0; j < N4; ++j) o4[j] = *reinterpret_cast<const f4_t *>(trow + off4(j, i16));
#pragma unroll
        for (int k = 0; k < N2; ++k) o2[k] = *reinterpret_cast<const f2_t *>(trow + off2(k, i16));
#pragma unroll
        for (int j = 0; j < N4; ++j) {
            const f4_t o = o4[j];
            f2_t olo = {o.x, o.y}, ohi = {o.z, o.w};
            if (LS) { olo += cb2; ohi += cb2; }
            const f2_t xlo = {a[j].x, a[j].y}, xhi = {a[j].z, a[j].w};
            const f2_t vlo = __builtin_elementwise_fma(xlo, r2, olo), vhi = __builtin_elementwise_fma(xhi, r2, ohi);
            const f4_t v = {vlo.x, vlo.y, vhi.x, vhi.y};
            if (!(kLast4 && j == N4 - 1) || col_ok) grad_store<NT>(reinterpret_cast<f4_t *>(g + off4(j, i16)), v);
        }
#pragma unroll
        for (int k = 0; k < N2; ++k) {
            f2_t o = o2[k];
            if (LS) o += cb2;
            const f2_t v = __builtin_elementwise_fma(c[k], r2, o);
            if (k < N2 - 1 || col_ok) grad_store<NT>(reinterpret_cast<f2_t *>(g + off2(k, i16)), v);
        }
    }
    // a "use" of every register of the row: where the compiler waits for the row's loads
    __device__ __forceinline__ void touch() const
    {
#pragma unroll
        for (int j = 0; j < N4; ++j) asm volatile("" ::"v"(a[j]));
#pragma unroll
        for (int k = 0; k < N2; ++k) asm volatile("" ::"v"(c[k]));
    }
    template <bool NT>
    __device__ static __forceinline__ void store_zero(float *g, int i16, bool col_ok)
    {
        const f4_t z4 = {0.f, 0.f, 0.f, 0.f};
        const f2_t z2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < N4; ++j)
            if (!(kLast4 && j == N4 - 1) || col_ok) grad_store<NT>(reinterpret_cast<f4_t *>(g + off4(j, i16)), z4);
#pragma unroll
        for (int k = 0; k < N2; ++k)
            if (k < N2 - 1 || col_ok) grad_store<NT>(reinterpret_cast<f2_t *>(g + off2(k, i16)), z2);
    }
};

// chunking of a row of C columns (C even): U = ceil(C / 32) units of 32 columns, taken as float4
// chunks of two units wherever the last chunk then still ends on a lane boundary
static bool r16_shape(int C, int &n4, int &n2)
{
    if (C < 2 || C > 256 || (C & 1)) return false;
    const int U = (C + 31) / 32;
    if (U & 1) { n4 = U / 2; n2 = 1; }
    else if ((C & 3) == 0) { n4 = U / 2; n2 = 0; }
    else { n4 = U / 2 - 1; n2 = 2; }
    return true;
}

// PS ("persistent", more samples than CUs): ONE workgroup per CU for the whole launch, taking the samples the
// one-sample-per-workgroup launch would have dispatched to it one after the other (virtual block vb = blockIdx +
// r * gridDim).  What it saves per sample is everything a fresh workgroup waits for before its chains can start:
// the dispatch, and above all the rows -- a worker holds the NEXT sample's rows in a second set of registers, loaded
// while this sample's chains run (instead of the 4-byte-per-line L2 prefetch of the one-sample form), so the next
// sample starts with its rows, lengths and labels already there.  Two workgroup barriers per sample (everyone is done
// with the lattice / the lattice is initialised); N4 / N2 whose two row sets do not fit 128 VGPRs keep the other form.
template <int N4, int N2, bool NT, bool PS = false>
__global__ __launch_bounds__(kThreads, 4) void noblank_r16_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    typedef R16Row<N4, N2> Row;
    constexpr int RP = Row::kCols;                           // floats per staged row
    constexpr int G = kPipeRows / 4;                         // groups of four rows per worker
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, RP);
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    int vb = blockIdx.x;                                     // (PS: advances by gridDim.x per sample)
    int b = xcd_sample(vb, p.B);
    if (tid == 64 * kChainB) note_arrival(p.counter, b);        // (the beta chain wave has no other vector-memory operation)
    const int u = (w == 0 || w == kChainB) ? -1 : (w < kChainB ? w - 1 : w - 2);
    const int rho = lane >> 4, i16 = lane & 15;              // row of the group, position inside the row

    if (CTC_DIAG(p) == 1) return;                                 // diagnostic: cost of the bare dispatch
    stamp(p, 0);
    stamp_setup(p, 0);
    auto spread = [&](int which) {                           // diagnostic (stop == -50): entry / exit times of
        if (CTC_DIAG(p) != -50 || w != 1 || lane != 0) return;    // the first, middle and last workgroup, wave 1
        const int bid = blockIdx.x, nb = gridDim.x;
        const int slot = bid == 0 ? 0 : bid == nb / 2 ? 2 : bid == nb - 1 ? 4 : -1;
        if (slot < 0) return;
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * (slot + which);
        o[0] = __builtin_amdgcn_s_memtime();
        o[1] = __builtin_amdgcn_s_memrealtime();
    };
    spread(0);
    // The two lengths come through the SCALAR memory path (uniform address): a vector load here is
    // followed by a readfirstlane, i.e. a full memory round trip BEFORE the first row load is issued,
    // and later waits on the row loads degrade to vmcnt(0).
    ScalarLengths len(p.in_len + b, p.tgt_len + b);
    // this lane's row in each group (the same rows t for every sample: they depend on T alone)
    int tv0[G];
    Row v[G];
    // PS: groups [0, NX) of the NEXT sample wait in a second set of registers (loaded while this sample's chains run:
    // group 0 is what the next sample's chains wait for), the other groups are loaded in place as soon as the
    // gradient pass is through with this sample's (P3 takes the groups last to first).
    // Measured, T = 150, C = 158, us per launch with NX = 1 / 2 / 3 (the one-sample form: 98.0 / 45.1 / 24.1):
    // B = 2048 (non-temporal stores) 82.9 / 82.7 / 90.1, B = 1024 (write-through stores) 47.0 / 42.8 / 41.7,
    // B = 512 26.4 / 24.9 / 24.3 -- NX = 0 (every group loaded in place: group 0's load queues behind the sample's last
    // gradient stores) 88.7 / 50.3 / 27.5.
#ifdef CTC_R16_NX
    constexpr int NX = PS ? CTC_R16_NX : 0;
#else
    constexpr int NX = PS ? (NT ? 2 : 3) : 0;
#endif
    Row nx[NX > 0 ? NX : 1];
    const bool col_ok = Row::off_last(i16) < p.C;            // last chunk inside the row
    const int c_last = col_ok ? Row::off_last(i16) : p.C - (Row::kLast4 ? 4 : 2);
    // Groups whose four slots all lie beyond the sequence (the last group of the last workers: 14 x 12 slots
    // for T rows) are skipped -- a group costs its instructions whatever its rows hold.  Wave-uniform.
    const int Hh = (p.T + 1) >> 1;
    bool grp[G];
#pragma unroll
    for (int g = 0; g < G; ++g) grp[g] = u >= 0 && 2 * kPipeWorkers * g + 2 * u < Hh;
    int klab = 0;                                            // PS, wave 0: the sample's raw labels (one per lane)
    if (PS && w == 0 && lane < p.S) klab = load_label(p.lab, p.lab64, (int64_t)b * p.S + lane);
    if (u >= 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tv0[g] = r16_row(p.T, u, g, rho);
            if (g >= NX) v[g].load(row_ptr(p, tv0[g] >= 0 ? tv0[g] : 0, b), i16, c_last);
            else nx[g].load(row_ptr(p, tv0[g] >= 0 ? tv0[g] : 0, b), i16, c_last);
        }
        stamp_setup(p, 1);                                   // loads issued
        if (PS) {
            // PS: every wait for a row load of the FIRST sample is taken here, once per launch (and in the same branch
            // as the loads: the compiler places its waits by paths, not by conditions).  Inside the loop a wait on a
            // load can only say "at most n operations outstanding", and a load that could still be in flight at the
            // loop's head would make the first groups' waits count the previous sample's gradient stores.
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g < NX) nx[g].touch();
                else v[g].touch();
            }
        }
    }
    const float ninf = -__builtin_inff();
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    // Per-sample state of the loops below (PS: one pass per sample of this workgroup; otherwise one pass).  The two
    // ROLES run loops of their own -- a wave never changes its role, and in one common loop everything a worker carries
    // from sample to sample (the next sample's rows ...) would stay allocated through the chains' code and vice versa.
    int Tb = 0, L = 0, bn = b, vbn = vb, klab_n = 0;
    bool more = false;                                       // (uniform over the workgroup)
    ScalarLengths len_n;
    const cell_t zero = make_cell(0.f, 0);
    // start of a sample: lengths, labels, the lattice's pads, the counters; ends in the workgroup's barrier
    auto begin = [&](const bool chain_role) -> bool {
        int64_t Tb64, L64;
        len.get(Tb64, L64);
        stamp_setup(p, 2);                                   // lengths there
        const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
        Tb = ok ? (int)Tb64 : 0;
        L = ok ? (int)L64 : 0;
        vbn = vb + (int)gridDim.x;
        more = PS && vbn < p.B;
        bn = more ? xcd_sample(vbn, p.B) : b;
        // PS: the next sample's lengths and labels travel while this one runs
        if (PS) len_n = ScalarLengths(p.in_len + bn, p.tgt_len + bn);
        klab_n = 0;
        // The labels are wave 0's business alone (S <= 31 < 64), in a branch of its own: a vector load on
        // the workers' (static) path would make every later wait on their row loads a vmcnt(0).
        if (chain_role && w == 0) {
            int k = 0;
            if (lane < L) {
                k = (PS ? klab : load_label(p.lab, p.lab64, (int64_t)b * p.S + lane)) % p.C;
                if (k < 0) k += p.C;                         // python negative index (NoBlankCTC.py:102)
            }
            if (lane < p.SP) sm.lab[lane] = k;
            if (more && lane < p.S) klab_n = load_label(p.lab, p.lab64, (int64_t)bn * p.S + lane);
        }
        stamp_setup(p, 3);                                   // (wave 0: labels stored)
        // (LDS initialisation that needs no loaded value: it overlaps the loads' latency)
        for (int i = tid; i < (p.SP + 1) * sm.TP; i += kThreads) sm.em[i - kR16Pad] = zero;   // pads + spare row
        if (tid < 16) sm.cnt[tid] = 0;
        if (tid == 16) *sm.done = 0;
        if (tid == 17) sm.occ[((p.SP + 3) & ~3) + 1] = 0;
        if (tid < 8) sm.dummy[tid] = 0.f;
        // LDS-only barrier: __syncthreads() would also wait for every row load of the wave (s_waitcnt
        // vmcnt(0)), but a worker only needs its first group's rows to start
        stamp_setup(p, 4);                                   // LDS initialised, at the barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stamp(p, 1);
        stamp_setup(p, 5);
        stamp_mode(p, -200, (vb - (int)blockIdx.x) / (int)gridDim.x);   // (per sample: the lattice is initialised)
        if ((vb - (int)blockIdx.x) / (int)gridDim.x < 6) stamp_mode(p, -400, (vb - (int)blockIdx.x) / (int)gridDim.x);   // (-400: starts in slots 0..5, ends in 6..11)
        return CTC_DIAG(p) != 2;                                  // diagnostic: dispatch + setup (+ loads in flight)
    };
    // end of a sample: false when it was the workgroup's last; otherwise on to the next one, behind a barrier
    // (every wave is done with this sample's lattice and tiles)
    auto advance = [&]() -> bool {
        stamp_mode(p, -300, (vb - (int)blockIdx.x) / (int)gridDim.x);   // (per sample: this wave is done with it)
        if ((vb - (int)blockIdx.x) / (int)gridDim.x < 6) stamp_mode(p, -400, 6 + (vb - (int)blockIdx.x) / (int)gridDim.x);
        if (!more) return false;
        vb = vbn;
        b = bn;
        len = len_n;
        klab = klab_n;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        return true;
    };
    // ---------------------------------------------------------------- chain waves
    auto chain_sample = [&](const NoblankParams &p, const int lane) {
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, RP);
    if (Tb == 0) {                                           // no alignment exists: nll = 1e13 (the workers: zero gradient)
        if (w == 0)
            publish_and_reduce_sum(-kNeg, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        return;
    }
    {
        __builtin_amdgcn_s_setprio(3);                       // the chains are the critical path
        if (w == 0) {
            const cell_t a = r16_chain<true>(p, sm, p.T, Tb, L, p.SP, lane);
            stamp(p, 11);
            // nll = -log alpha[T_b-1, L_b-1] (NoBlankCTC.py:58-68,139).  The chains ran on emissions that lack a
            // per-row constant (the row's log-sum-exp and, smoothed, the b sum_n lp[n] term: the posteriors are
            // normalised per row and do not see it); the workers left the sums of those constants over their live
            // rows in sm.cs -- 56 values, added here in a fixed order (bitwise reproducible).
            bool late = false;
            {
                int spins = 0;
                while (*(lds_cvint *)sm.done < kPipeWorkers) {
                    if (++spins >= kSpinLimit) { late = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                lds_order();
            }
            const float csum = wave_sum(lane < 4 * kPipeWorkers ? sm.cs[lane] : 0.f);
            const float am = a.x;
            float nll = am > 0.f ? -(__builtin_amdgcn_logf(am) + (float)(cell_k(a) - kXrBias) + csum) * kLn2 : -kNeg;
            if (am != am) nll = am;                          // starved hand-off: NaN, not a plausible number
            if (late) {
                raise_status(p.counter, kStatusNoblankStarved);
                nll = __builtin_nanf("");
            }
            publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
            if (CTC_DIAG(p) == -50 && lane == 0) {                // diagnostic: when the alpha wave (loss ticket) is done
                const int bid = blockIdx.x, nb = gridDim.x;
                const int slot = bid == 0 ? 6 : bid == nb / 2 ? 7 : bid == nb - 1 ? 8 : -1;
                if (slot >= 0) {
                    unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * slot;
                    o[0] = __builtin_amdgcn_s_memtime();
                    o[1] = __builtin_amdgcn_s_memrealtime();
                }
            }
        } else if (p.grad || p.gamma) {
            r16_chain<false>(p, sm, p.T, Tb, L, p.SP, lane);
            stamp(p, 11);
        }
    }
    };                                                       // (chain_sample)

    // ---------------------------------------------------------------- workers
    // (`lane`, `u`: opaque copies of the outer values, new ones per sample.  Nearly everything a worker computes --
    // tile and cell addresses, row pointers, masks -- derives from the lane, the worker index and launch constants;
    // in the PS loop the compiler would hoist ALL of it out of the loop and keep it in registers: 128 VGPRs and
    // a hundred spilled SGPRs for rows of two registers.)
    auto worker_sample = [&](const NoblankParams &p, const int lane, const int u) {
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, RP);
    const int rho = lane >> 4, i16 = lane & 15;
    int tv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) tv[g] = PS ? opaque_v(tv0[g]) : tv0[g];
    // PS launches always want the gradient and never the posteriors (the host takes the one-sample form for those), and
    // a sample without alignment goes through the ordinary path as a sample without live rows (T_b = L_b = 0: nothing
    // is published, every scale is 0, the gradient rows come out as zeros): ONE path through the loop.  The compiler
    // places its waits for the next sample's row loads by control-flow paths; with a side path that loads and leaves,
    // the main path waited for its own last gradient stores at the end of every sample.
    const bool has_grad = PS || p.grad != nullptr, has_gamma = !PS && p.gamma != nullptr;
    // PS: the next sample's rows into the spare register set (all of this sample's rows are in `v` by then)
    // (unconditional, like every load of a next sample's rows: behind the workgroup's last sample they fetch that
    // sample's rows once more, and nobody waits for them -- a load under a condition leaves the compiler with two
    // candidates for the registers at the loop's head, and it keeps both: 30 registers and as many copies)
    auto load_next = [&]() {
        if (!PS) return;
#pragma unroll
        for (int g = 0; g < NX; ++g) nx[g].load(row_ptr(p, tv[g] >= 0 ? tv[g] : 0, bn), i16, c_last);
    };
    // ... and of the groups that are loaded in place, once this sample no longer needs the registers
#define CTC_R16_RELOAD(g)                                                                             \
    do {                                                                                              \
        if (PS && (g) >= NX) v[g].load(row_ptr(p, tv[g] >= 0 ? tv[g] : 0, bn), i16, c_last); \
    } while (0)
    if (!PS && Tb == 0) {                                    // no alignment exists: zero gradient
        if (p.grad) {
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (tv[g] >= 0) Row::template store_zero<NT>(p.grad + ((int64_t)tv[g] * p.B + b) * p.C, i16, col_ok);
        }
        if (p.gamma) {                                       // posteriors of a sample without alignment: zeros
#pragma unroll
            for (int g = 0; g < G; ++g)
                for (int l = i16; l < p.S && tv[g] >= 0; l += 16) p.gamma[((int64_t)b * p.T + tv[g]) * p.S + l] = 0.f;
        }
        return;
    }
    float *tile = sm.stage + (size_t)u * 4 * RP;             // this worker's staging / occupancy tile
    float *trow = tile + rho * RP;                           // this lane's row of it
    // states served by this lane in the two passes: l = i16 and l = 16 + i16 (S <= 31)
    int lst[2];
    float *gat[2];                                           // tile address of class lab[l] in this lane's row
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        lst[s] = 16 * s + i16;
        gat[s] = trow + (lst[s] < p.SP ? sm.lab[lst[s]] : 0);
    }
    const bool own[2] = {lst[0] < L, lst[1] < L};
    const bool smooth = p.ls_b != 0.f;                       // (wave-uniform)
    cell_t *const spare_w = reinterpret_cast<cell_t *>(sm.dummy);
    float mrow[G];                                           // row maximum
    // P1a -- what the CHAINS wait for, for all groups first: the labels' logits relative to the row maximum,
    // e = (x[lab_l] - max) log2e (times a when smoothed), split into 2^floor * 2^frac cells.  The row's
    // log-sum-exp is a per-row constant that the posteriors never see (they are normalised per row) and the
    // loss takes as a sum (P1b, below): the 158 exponentials of a row are no longer in front of the chains.
#pragma unroll
    for (int g = 0; g < G; ++g) {
        // A SIMD serves its waves oldest first, and the chains cannot start before EVERY worker has
        // published its first group: the waves that are behind go first.
        if (g == 0) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(1);
        if (g < NX) v[g] = nx[g];
        if (grp[g]) {
            Row &x = v[g];
            const int t = tv[g];
            const bool live = t >= 0 && t < Tb;
            float m = x.max();
            row16_allmax(m);
            mrow[g] = m;
            if (CTC_DIAG(p) < 0 && g == 0) stamp(p, 6);           // diagnostic: the first group's rows are there
            // raw rows -> tile, labels' logits back
            x.to_tile(trow, i16);
            lds_order();
            float xv[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) xv[s] = *gat[s];
            lds_order();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float ec = (xv[s] - m) * kLog2e;
                const float e2 = fmaxf(smooth ? p.ls_a * ec : ec, kXrMinLog2);
                const float fl = __builtin_floorf(e2);
                const float pm = __builtin_amdgcn_exp2f(e2 - fl);
                cell_t *dst = (live && lst[s] < p.SP) ? sm.em + lst[s] * sm.TP + t : spare_w;
                *dst = own[s] ? make_cell(pm, (int)fl) : zero;
            }
            lds_order();
        } else {
            mrow[g] = 0.f;
        }
        sm.cnt[u] = 4 * (g + 1);                             // publishes the four slots (same wave: in order)
        // nothing of the next group may be scheduled in front of this publication: the chains wait
        // for it, and the next group's first instruction waits for loads that are still in flight
        __builtin_amdgcn_sched_barrier(0);
        if (CTC_DIAG(p) < 0) stamp(p, 8 + g);                     // diagnostic: group g published
    }
    __builtin_amdgcn_s_setprio(0);
    load_next();
    // P1b -- beside the running chains: the row registers become exp(x - max) (P3 needs softmax(x) = that times
    // 1/sum), the per-row constant of the loss is  -a log2 sum + b sum_n lp[n]  in log2 units
    // (sum_n lp[n] = (sum_n x_n - C max) log2e - C log2 sum, NoBlankCTC.py:100-107; plain loss: a = 1, b = 0).
    float rs[G];                                             // grad_scale / sum_c exp(x - max), 0 for dead rows
    float cacc = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        rs[g] = 0.f;
        if (!grp[g]) continue;
        Row &x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tb;
        const float m = mrow[g];
        float sx = 0.f;                                      // label smoothing: sum of the row's logits
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
    // Occurrence index of every state among equal labels (0 = first): repeated labels add up in the workers' occupancy
    // tiles, one plain read-modify-write pass per repetition (P3).  The last worker makes the table here, while the
    // chains run -- it used to sit in front of the alpha chain, which therefore started (and ended) 0.3 us behind beta.
    if (u == kPipeWorkers - 1 && has_grad) {
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
        sm.occ[((p.SP + 3) & ~3) + 1] = 1;                   // the table is there
    }
    // More samples than CUs: the workgroup that follows this one on the CU (dispatch order: block + one full
    // round of CUs, the same XCD under round-robin placement -- speed only) will want the rows of ITS sample.
    // One 4-byte load per lane, one 128-byte line each, pulls that sample into this XCD's L2 while the chains
    // of this one run: 14 workers x 64 lanes cover its T rows of C floats.  The value is never used.
    float prefetched = 0.f;
    if (!PS && p.next_round > 0 && (int)blockIdx.x + p.next_round < p.B) {
        const int nb = xcd_sample(blockIdx.x + p.next_round, p.B);
        const int line = u * kWave + lane;                   // 0 .. 895
        const int lines_per_row = (p.C * 4 + 127) / 128;
        const int t = line / lines_per_row, c = (line - t * lines_per_row) * 32;
        // (plain load through the compiler: it keeps the destination register reserved until the value is
        // there; the add below is the "use", placed where the load has long returned)
        typedef const float __attribute__((address_space(1))) gfloat;
        if (t < p.T) prefetched = *(gfloat *)(row_ptr(p, t, nb) + (c < p.C ? c : p.C - 1));
    }
    if (!has_grad && !has_gamma) return;

    const int Tlive = Tb;
    const float gsc = p.grad_scale;
    Row::zero_tile(trow, i16);                               // the occupancy tile starts all zero and every group leaves it so
    lds_order();
    if (PS) {
        // the next sample's rows are waited for HERE, in the workers' idle window and with no gradient store in flight
        // (see the launch's first wait above)
#pragma unroll
        for (int g = 0; g < NX; ++g) nx[g].touch();
    }
    // how far the chains must have come for each group (scalars; made here, in the workers' idle window, so that the
    // gradient pass behind the chains carries as little scalar bookkeeping as possible -- a wave issues ONE instruction
    // at a time, scalar or vector)
    int need_a_[G], need_b_[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        need_a_[g] = 0;
        need_b_[g] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tk = r16_row(p.T, u, g, k);            // scalar twin of tv[g]
            if (tk >= 0 && tk < Tlive) {
                need_a_[g] = max(need_a_[g], tk + 1);
                need_b_[g] = max(need_b_[g], Tlive - tk);
            }
        }
    }
    // ... and the addresses the gradient pass needs -- the lattice row of the lane's two states, the gradient row of every
    // group: 32-bit multiplies (quarter rate) and 64-bit address arithmetic that the compiler otherwise redoes per group,
    // behind the chains (made opaque: a value it cannot see through stays in its register)
    int crow[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) crow[s] = opaque_v((lst[s] < p.SP ? lst[s] : p.SP) * sm.TP);
    float *grow[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        grow[g] = p.grad + ((int64_t)(tv[g] > 0 ? tv[g] : 0) * p.B + b) * p.C;
        asm volatile("" : "+v"(grow[g]));
    }
    bool starved = false;

    // P3: middle-out, one look at the chains' progress per group
    // Every live row has the SAME total sum_l alpha_t(l) beta_t(l) (the sample's likelihood), so the exponent
    // that brings a row's terms into range need not be the row's own maximum: a lane keeps the one of the first
    // live row it meets (`kref`) and the later groups skip that reduction.
    int kref = 0;
    bool have_ref = false;
    // the lane's two states in the occurrence table (made by the last worker right behind its P1b, long before the chains
    // cross): read once per sample, here, not once per group behind the chains
    int occn[2] = {0, 0}, max_occ = 0;
    if (has_grad) {
        int spins = 0;
        while (*(lds_cvint *)(sm.occ + ((p.SP + 3) & ~3) + 1) == 0) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        lds_order();
        occn[0] = own[0] ? sm.occ[lst[0]] : 0;
        occn[1] = own[1] ? sm.occ[lst[1]] : 0;
        max_occ = __builtin_amdgcn_readfirstlane(sm.occ[(p.SP + 3) & ~3]);
    }
    const bool first_occ[2] = {own[0] && occn[0] == 0, own[1] && occn[1] == 0};
    float *const sink = sm.dummy + 4;                        // (write-only)
#pragma unroll
    for (int g = G - 1; g >= 0; --g) {
        if (!grp[g]) continue;
        const int need_a = need_a_[g], need_b = need_b_[g];
        if (need_a > 0) {
            int spins = 0;
            while (*(lds_cvint *)(sm.cnt + kPipeWorkers) < need_a || *(lds_cvint *)(sm.cnt + kPipeWorkers + 1) < need_b) {
                if (++spins >= kSpinLimit) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();
        }
#ifdef CTC_AMD_FAULT_INJECT                                  // tests/test_status.py: sample 1's workers "starve"
        if (b == 1) starved = true;
#endif
        if (CTC_DIAG(p) < 0) stamp(p, 3 + (G - 1 - g));
        if (g == G - 1) asm volatile("" ::"v"(prefetched));    // the prefetched value is "used" here (and dropped)
        const Row &x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tlive;
        // gamma_t(l) = alpha_t(l) beta_t(l) / sum_l' (...): mantissa products, exponents added and
        // shifted by the reference exponent
        // (beta comes with the emission of its own row: q = beta p, see r16_chain)
        float pr[2];
        int ks[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool in = live && own[s];
            const int off = crow[s] + (t > 0 ? t : 0);       // a valid address, masked afterwards
            const cell_t a = sm.al[off], bb = sm.be[off], e = sm.em[off];
            pr[s] = in ? a.x * bb.x * __builtin_amdgcn_rcpf(e.x) : 0.f;
            ks[s] = cell_k(a) + cell_k(bb) - cell_k(e);
        }
        if (__builtin_amdgcn_ballot_w64(live && !have_ref) != 0) {   // (wave-uniform branch)
            int km = max(pr[0] > 0.f ? ks[0] : 0, pr[1] > 0.f ? ks[1] : 0);
            row16_allmax(km);
            if (live && !have_ref) { kref = km; have_ref = true; }
        }
        float z[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) z[s] = __builtin_amdgcn_ldexpf(pr[s], ks[s] - kref);
        float tot = z[0] + z[1];
        row16_allsum(tot);
        if (has_gamma) {                                     // (wave-uniform) ctc_amd_noblank_posteriors: gamma[b][t][l] instead of a gradient
            const float ginv = (live && tot > 0.f) ? __builtin_amdgcn_rcpf(tot) : 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (t >= 0 && lst[s] < p.S) p.gamma[((int64_t)b * p.T + t) * p.S + lst[s]] = own[s] ? z[s] * ginv : 0.f;
            continue;
        }
        // (smoothed: grad = (1 - b) softmax - a occupancy - b, all times 1/B on live rows)
        float rinv = (live && tot > 0.f) ? -gsc * p.ls_a * __builtin_amdgcn_rcpf(tot) : 0.f;   // (the tile holds MINUS the occupancy)
        if (starved) {                                       // never observed; loud if a hand-off were broken
            rinv = __builtin_nanf("");
            raise_status(p.counter, kStatusNoblankStarved);
        }
        // MINUS the class occupancy of the four rows, scattered into the all-zero tile
        // (only lanes that own a state write; pass k adds the k-th repetition of a label)
        // (through an address select, not under an exec mask: a predicated store costs the wave a compare, a mask
        // save, the store and a mask restore -- one instruction at a time, scalar or vector; lanes without a state write
        // a spare word)
#pragma unroll
        for (int s = 0; s < 2; ++s) *(first_occ[s] ? gat[s] : sink) = z[s] * rinv;
        for (int k = 1; k <= max_occ; ++k) {
            lds_order();
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (own[s] && occn[s] == k) *gat[s] += z[s] * rinv;
        }
        lds_order();
        // dense rows: grad = softmax(x) * scale - occupancy   (dead rows: scale = occupancy = 0)
        if (t >= 0) {
            float *gp = grow[g];
            if (smooth) x.template store_grad<NT, true>(gp, trow, i16, rs[g] * (1.f - p.ls_b), col_ok, live ? -p.ls_b * gsc : 0.f);
            else x.template store_grad<NT, false>(gp, trow, i16, rs[g], col_ok, 0.f);
        }
        lds_order();
        // the tile is all zeros again: the label slots, not three 16-byte stores per lane (an LDS store costs the wave
        // issue time by the byte: tools/micro/km_probe.hip)
#pragma unroll
        for (int s = 0; s < 2; ++s) *(own[s] ? gat[s] : sink) = 0.f;
        lds_order();
        CTC_R16_RELOAD(g);
    }
    stamp(p, 7);
    spread(1);
#undef CTC_R16_RELOAD
    };                                                       // (worker_sample)

    if (u < 0) {
        for (;;) {
            if (!begin(true)) return;
            chain_sample(p, PS ? opaque_v(lane) : lane);
            if (!advance()) return;
        }
    }
    for (;;) {
        if (!begin(false)) return;
        if (PS) worker_sample(p, opaque_v(lane), opaque_s(u));
        else worker_sample(p, lane, u);
        if (!advance()) return;
    }
}

// ---- diagnostic probe (tools/chain_probe.py): the chains alone, every row already published ------
// (diagnostics build only: the product library carries no kernel it never launches)
#ifdef CTC_AMD_DIAGNOSTICS
// `mode`: what the other waves do while the chains run -- 0: poll the chain's progress and sleep (a waiting
// worker); 1: VALU only (fma + one exp in eight, like a worker's P1 / P3 arithmetic); 2: LDS only (16-byte tile
// writes and 8-byte cell reads in the wave's own staging tile); 3: both, in a worker's proportions (150 : 10);
// +8: at wave priority 2 (a worker's first group).  The chains are waves 0 and `chain_b`.
__global__ __launch_bounds__(kThreads, 4) void r16_chain_probe_kernel(NoblankParams p, unsigned long long *out, int waves_alive,
                                                                     int mode, int chain_b)
{
    extern __shared__ float4 smem_raw[];
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, 160);
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    for (int i = tid; i < 3 * (p.SP + 1) * sm.TP; i += kThreads) sm.em[i - kR16Pad] = make_cell(1.5f, -1);
    if (tid < 16) sm.cnt[tid] = kPipeRows;                   // everything published
    __syncthreads();
    if (w >= waves_alive && w != chain_b) return;
    if (w != 0 && w != chain_b) {                            // bystanders
        typedef const volatile __attribute__((address_space(3))) int lds_cvint;
        int spins = 0;
        const int u = (w < chain_b ? w - 1 : w - 2) % kPipeWorkers;
        float *tile = sm.stage + (size_t)u * 4 * 160 + (lane >> 4) * 160 + 4 * (lane & 15);
        const cell_t *cells = sm.al + (size_t)(lane & 15) * sm.TP;
        if (mode & 8) __builtin_amdgcn_s_setprio(2);
        float a0 = lane, a1 = 1.f, a2 = 2.f, a3 = 3.f, acc = 0.f;
        const int kind = mode & 7;
        while (*(lds_cvint *)(sm.cnt + kPipeWorkers) < p.T && ++spins < 100000) {
            if (kind == 0) { __builtin_amdgcn_s_sleep(8); continue; }
            if (kind & 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {               // 16 x (7 fma + 1 exp) = 128 VALU
                    a0 = __builtin_fmaf(a0, 1.0001f, a1); a1 = __builtin_fmaf(a1, 0.9999f, a2);
                    a2 = __builtin_fmaf(a2, 1.0001f, a3); a3 = __builtin_fmaf(a3, 0.9999f, a0);
                    a0 = __builtin_fmaf(a0, 0.5f, a2); a1 = __builtin_fmaf(a1, 0.5f, a3);
                    a2 = __builtin_fmaf(a2, 0.5f, 1.f);
                    a3 = __builtin_amdgcn_exp2f(a3 * 1e-3f);
                }
            }
            if (kind & 2) {
                const int reps = kind == 2 ? 12 : 1;
                for (int r = 0; r < reps; ++r) {             // per 128 VALU: 3 x 16-byte + 2 x 16-byte tile traffic, 6 cell reads
                    f4_t v4 = {a0, a1, a2, a3};
                    *reinterpret_cast<f4_t *>(tile) = v4;
                    *reinterpret_cast<f4_t *>(tile + 64) = v4;
                    *reinterpret_cast<f2_t *>(tile + 128) = f2_t{a0, a1};
                    lds_order();
                    const f4_t r4 = *reinterpret_cast<const f4_t *>(tile + 64);
                    const cell_t c0 = cells[8 + (spins & 63)], c1 = cells[sm.TP * 16 + 9 + (spins & 63)];
                    acc += r4.x + c0.x + c1.x;
                    lds_order();
                }
            }
        }
        if (acc + a0 + a1 + a2 + a3 == 12345.f) out[15] = 1;  // (keeps the loop alive)
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const cell_t c = w == 0 ? r16_chain<true>(p, sm, p.T, p.T, p.SP, p.SP, lane) : r16_chain<false>(p, sm, p.T, p.T, p.SP, p.SP, lane);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) {
        out[w == 0 ? 0 : 1] = t1 - t0;
        out[4 + (w == 0 ? 0 : 1)] = (unsigned long long)cell_k(c);
    }
}
#endif  // CTC_AMD_DIAGNOSTICS

}  // namespace ctc
