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
//     scaled posteriors are scattered into it with LDS float atomics (ds_add_f32, repeated
//     labels simply add up -- no class tables, no duplicate folding), and read back 8 bytes per
//     lane next to the resident exp(x - max) values.
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
// LDS access (the chain is bound by its LDS instructions, not by its arithmetic): the pad is odd so
// that the forward pairs (steps 1-2, 3-4, ...) start on 16-byte boundaries; beta_t is stored one
// cell further (at kR16Pad + 1 + t), which makes a beta step read and write the SAME cell index
// (it reads the emissions of row t+1 and writes row t), its pairs aligned whenever the alpha pairs
// of that parity are (one single step is peeled off when T_b is even).  Pitch TP = 2 (mod 4)
// cells: 16-byte aligned rows, and the lanes of a chain access (4 banks each, stride 2 TP words)
// fall on different banks.  kR16Pad cells before t = 0 and after t = T-1 of every state row
// take the chains' prefetch (4 steps ahead) and the last, partly idle group of four steps (up to 3
// steps past the end); `em` pads are zero.  One spare state row SP: all zeros in `em` (read by
// the chain lanes that only watch progress counters), scratch in `al` / `be`.
constexpr int kR16Pad = 2 * kPrefetch + 1;                  // odd
__host__ __device__ inline int r16_pitch(int T)
{
    int tp = T + 2 * kR16Pad + 1;
    while ((tp & 3) != 2) ++tp;
    return tp;
}

struct R16Smem {
    int TP;
    cell_t *em, *al, *be;                                    // (mantissa, exponent) cells, [state][time]
    float *dummy, *stage;
    int *cnt, *lab, *occ;
    __device__ R16Smem(float *base, int T, int SP, int RP)
    {
        TP = r16_pitch(T);
        cell_t *lat = reinterpret_cast<cell_t *>(base);
        em = lat + kR16Pad;                                  // -> cell (t = 0, l = 0)
        al = em + (size_t)(SP + 1) * TP;
        be = al + (size_t)(SP + 1) * TP + 1;                 // beta_t one cell further (see above)
        dummy = reinterpret_cast<float *>(lat + (size_t)3 * (SP + 1) * TP);   // write-only spare cells
        cnt = reinterpret_cast<int *>(dummy + 8);
        lab = cnt + 16;
        occ = lab + ((SP + 3) & ~3);                        // [SP] occurrence index, [SPpad] their maximum
        stage = reinterpret_cast<float *>(occ + ((SP + 3) & ~3) + 4);   // [workers][4 rows][RP]
    }
};

static size_t r16_smem_bytes(int T, int SP, int C)
{
    const int RP = 32 * ((C + 31) / 32);
    return (size_t)3 * (SP + 1) * r16_pitch(T) * 8 + (8 + 16 + 2 * ((SP + 3) & ~3) + 4) * 4 +
           (size_t)kPipeWorkers * 4 * RP * 4;
}

// alpha (FWD) / beta (!FWD) chain of this kernel: (mantissa, exponent) cells as in xr_chain_sync
// (noblank_xr.hpp), alpha_t stored with, beta_t without the emission of step t.
// Hand-off: a worker publishes its rows in groups of four slots; group g of ALL workers together
// covers positions [28 g, 28 g + 28) of each half of the sequence, so "rows up to position q are
// there" is one scalar: the number of groups every worker must have finished.  It changes three
// times per chain; a block of steps whose requirement is already met costs no look at all.
// Returns alpha[T_b-1, L_b-1] (FWD) as a cell.
template <bool FWD>
__device__ __forceinline__ cell_t r16_chain(const NoblankParams &p, const R16Smem &sm, int T, int Tb, int L, int SP)
{
    constexpr int G = kPipeRows / 4;
    constexpr int D = FWD ? 1 : -1;
    const int lane = lane_id();
    // lanes beyond the states (and beyond the 14 progress counters) sit the chain out: a DPP read
    // from a disabled lane is the zero the first / last state needs anyway (bound_ctrl)
    if (lane >= (SP > kPipeWorkers ? SP : kPipeWorkers)) return make_cell(0.f, 0);
    const int lrow = lane < SP ? lane : SP;                  // counter-only lanes: the spare state row
    // step i = 1..T_b-1 reads the emissions of row tr(i) = FWD ? i : T_b - i and writes row
    // FWD ? i : T_b - 1 - i -- both at cell index x(i) = FWD ? i : T_b - i of their arrays
    const cell_t *erow = sm.em + (size_t)lrow * sm.TP;
    cell_t *orow = (FWD ? sm.al : sm.be - 1) + (size_t)lrow * sm.TP;   // (be - 1: index x, not t)
    typedef float pair_t __attribute__((ext_vector_type(4)));           // two cells: (lower, higher) time index
    float m;
    int k;

    const int H = (T + 1) >> 1;
    const int pos0 = FWD ? 0 : T - Tb;                       // position (in this chain's half order) of step 0
    const int own = FWD ? H : T - H;
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    const int *cp = sm.cnt + (lane < kPipeWorkers ? lane : 0);
    int have = 0;                                            // groups known to be published by every worker
    auto wait_upto = [&](int i_last) {                       // rows of steps 0..i_last must be published
        const int q = pos0 + (i_last < Tb ? i_last : Tb - 1);
        int ng = q / (2 * kPipeWorkers) + 1;
        if (q >= own || ng > G) ng = G;
        if (ng <= have) return;                              // (scalar) nothing new to wait for
        const int need = lane < kPipeWorkers ? 4 * ng : 0;
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(*(lds_cvint *)cp < need) != 0 && ++spins < kSpinLimit)
            __builtin_amdgcn_s_sleep(1);
        lds_order();
        have = ng;
    };
    auto merge = [&]() {                                     // (m,k) += neighbour, in the larger exponent
        const int nk = xr_nb<FWD>(k);
        const float nm = __builtin_bit_cast(float, xr_nb<FWD>(__builtin_bit_cast(int, m)));
        const int kk = k > nk ? k : nk;
        m = __builtin_amdgcn_ldexpf(m, k - kk) + __builtin_amdgcn_ldexpf(nm, nk - kk);
        k = kk;
    };
    auto step = [&](float em_, float ek_, bool norm) {       // one time step with emission (em_, ek_)
        if (FWD) merge();
        m *= em_;
        k += __builtin_bit_cast(int, ek_);
        if (norm) {                                          // mantissa back into [0.5, 1)
            k += __builtin_amdgcn_frexp_expf(m);
            m = __builtin_amdgcn_frexp_mantf(m);
        }
        if (!FWD) merge();
    };
    // steps (i, i+1) with the pair of emission cells `e` (lower, higher index); returns the pair to store
    auto step2 = [&](pair_t e, bool norm_second) {
        pair_t o;
        step(FWD ? e.x : e.z, FWD ? e.y : e.w, false);
        const float m1 = m;
        const int k1 = k;
        step(FWD ? e.z : e.x, FWD ? e.w : e.y, norm_second);
        const float k1f = __builtin_bit_cast(float, k1), k2f = __builtin_bit_cast(float, k);
        o.x = FWD ? m1 : m;  o.y = FWD ? k1f : k2f;          // lower index: the earlier step when walking up
        o.z = FWD ? m : m1;  o.w = FWD ? k2f : k1f;
        return o;
    };

    int *prog = sm.cnt + kPipeWorkers + (FWD ? 0 : 1);
    __builtin_amdgcn_s_setprio(3);                           // the chains are the critical path
    wait_upto(kPrefetch + 1);
    int i = 1;
    if (FWD) {                                               // alpha_0 = p_0(0) on state 0 only
        const cell_t e0 = erow[0];
        m = lane == 0 ? e0.x : 0.f;
        k = lane == 0 ? kXrBias + cell_k(e0) : 0;
        orow[0] = make_cell(m, k);
    } else {                                                 // beta_{T_b-1} = 1 on state L-1 only
        m = lane == L - 1 ? 1.f : 0.f;
        k = lane == L - 1 ? kXrBias : 0;
        orow[Tb] = make_cell(m, k);
        if ((Tb & 1) == 0 && Tb > 1) {                       // align the pairs: one single step (wave-uniform)
            const cell_t e = erow[Tb - 1];
            step(e.x, e.y, false);
            orow[Tb - 1] = make_cell(m, k);
            i = 2;
        }
    }
    // x(i) of the current step, and the lowest cell index of the pair (i, i+1): even by construction
    const int x0 = FWD ? i : Tb - i;
    constexpr int kPairs = kBlockSteps / 2;
    // block-relative bases at the LOWEST address a block touches, so that the unrolled pairs use
    // non-negative immediate offsets in both directions (ds offsets are unsigned).  The reads run
    // kPrefetch steps (two pairs) ahead of the writes.
    const pair_t *rb = reinterpret_cast<const pair_t *>(erow + (FWD ? x0 + kPrefetch : x0 - kPrefetch - (kBlockSteps - 1)));
    pair_t *wb = reinterpret_cast<pair_t *>(orow + (FWD ? x0 : x0 - (kBlockSteps - 1)));
    pair_t ring[2];
    {
        const pair_t *r0 = reinterpret_cast<const pair_t *>(erow + (FWD ? x0 : x0 - 3));
        ring[0] = r0[FWD ? 0 : 1];                           // steps i, i+1
        ring[1] = r0[FWD ? 1 : 0];                           // steps i+2, i+3
    }
    for (; i + kBlockSteps <= Tb; i += kBlockSteps) {
        lds_order();
        *prog = i;                                           // steps < i are done (every lane, same value)
        if (have < G) wait_upto(i + kBlockSteps - 1 + kPrefetch);
        if (p.stop < 0) stamp(p, 2 + i / kBlockSteps);       // diagnostic: block starts -> slots 2..10
#pragma unroll
        for (int q = 0; q < kPairs; ++q) {
            const pair_t e = ring[q & 1];
            ring[q & 1] = rb[FWD ? q : kPairs - 1 - q];
            wb[FWD ? q : kPairs - 1 - q] = step2(e, (q & 1) == 1);
        }
        rb += D * kPairs;
        wb += D * kPairs;
    }
    lds_order();
    *prog = i;
    if (have < G) wait_upto(Tb - 1);
    // the rest in groups of kPrefetch steps = one revolution of the ring (no guards inside).  The
    // last group may run up to three steps past the end: those read zero pad cells and write pad
    // cells of the output rows, which nobody looks at -- the final state is read back below.
    static_assert(kPrefetch == 4, "tail groups assume one ring revolution = one renormalisation period");
    if (!FWD) { rb += kPairs - 2; wb += kPairs - 2; }        // lowest address of a 4-step group
    for (; i < Tb; i += kPrefetch) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const pair_t e = ring[q];
            ring[q] = rb[FWD ? q : 1 - q];
            wb[FWD ? q : 1 - q] = step2(e, q == 1);
        }
        rb += D * 2;
        wb += D * 2;
    }
    lds_order();
    *prog = Tb;
    __builtin_amdgcn_s_setprio(0);
    // alpha[T_b-1, L_b-1] from where the chain stored it (the registers may hold overrun steps)
    return (FWD ? sm.al : sm.be)[(size_t)(L - 1) * sm.TP + (FWD ? Tb - 1 : 0)];
}

template <int CH2>
__global__ __launch_bounds__(kThreads, 4) void noblank_r16_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    constexpr int RP = 32 * CH2;                             // floats per staged row
    constexpr int G = kPipeRows / 4;                         // groups of four rows per worker
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, RP);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    const int u = (w == 0 || w == kChainB) ? -1 : (w < kChainB ? w - 1 : w - 2);
    const int rho = lane >> 4, i16 = lane & 15;              // row of the group, position inside the row
    const float ninf = -__builtin_inff();

    if (p.stop == 1) return;                                 // diagnostic: cost of the bare dispatch
    stamp(p, 0);
    auto spread = [&](int which) {                           // diagnostic (stop == -50): entry / exit times of
        if (p.stop != -50 || w != 1 || lane != 0) return;    // the first, middle and last workgroup, wave 1
        const int bid = blockIdx.x, nb = gridDim.x;
        const int slot = bid == 0 ? 0 : bid == nb / 2 ? 2 : bid == nb - 1 ? 4 : -1;
        if (slot < 0) return;
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * (slot + which);
        o[0] = __builtin_amdgcn_s_memtime();
        o[1] = __builtin_amdgcn_s_memrealtime();
    };
    spread(0);
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const int raw_label = tid < p.S ? load_label(p.lab, p.lab64, (int64_t)b * p.S + tid) : 0;
    // this lane's row in each group, its columns: pairs (32j + 2i, 32j + 2i + 1)
    int tv[G];
    f2_t v[G][CH2];
    const int c_lane = 2 * i16;
    const bool col_ok = 32 * (CH2 - 1) + c_lane < p.C;       // last pair inside the row (C is even)
    const int c_last = col_ok ? 32 * (CH2 - 1) + c_lane : p.C - 2;
    if (u >= 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tv[g] = pipe_row(p.T, u, 4 * g + rho);
            const float *row = row_ptr(p, tv[g] >= 0 ? tv[g] : 0, b);
#pragma unroll
            for (int j = 0; j < CH2; ++j)
                v[g][j] = *reinterpret_cast<const f2_t *>(row + (j < CH2 - 1 ? 32 * j + c_lane : c_last));
        }
    }
    // (LDS initialisation that needs no loaded value goes first: it overlaps the loads' latency)
    const cell_t zero = make_cell(0.f, 0);
    for (int i = tid; i < (p.SP + 1) * sm.TP; i += kThreads) sm.em[i - kR16Pad] = zero;   // pads + spare row
    if (tid < 16) sm.cnt[tid] = 0;
    if (tid < 8) sm.dummy[tid] = 0.f;
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;
    if (tid < p.SP) {
        int k = 0;
        if (tid < L) {
            k = raw_label % p.C;
            if (k < 0) k += p.C;                             // python negative index (NoBlankCTC.py:102)
        }
        sm.lab[tid] = k;
    }
    // LDS-only barrier: __syncthreads() would also wait for every row load of the wave (s_waitcnt
    // vmcnt(0)), but a worker only needs its first group's rows to start
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stamp(p, 1);
    if (p.stop == 2) return;                                 // diagnostic: dispatch + setup (+ loads in flight)

    if (Tb == 0) {                                           // no alignment exists: nll = 1e13, zero gradient
        if (w == 0)
            publish_and_reduce(-kNeg, b, p.B, p.nll, p.loss, p.loss_scale, p.counter, [](float x, int) { return x; });
        if (u >= 0 && p.grad) {
            const f2_t zero2 = {0.f, 0.f};
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (tv[g] < 0) continue;
                float *gp = p.grad + ((int64_t)tv[g] * p.B + b) * p.C + c_lane;
#pragma unroll
                for (int j = 0; j < CH2; ++j)
                    if (j < CH2 - 1 || col_ok) __builtin_nontemporal_store(zero2, reinterpret_cast<f2_t *>(gp + 32 * j));
            }
        }
        return;
    }
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;

    // ---------------------------------------------------------------- chain waves
    if (u < 0) {
        if (w == 0) {
            // while the first rows are on their way: occurrence index of every state among equal
            // labels (0 = first).  Repeated labels add up in the workers' occupancy tiles, one
            // plain read-modify-write pass per repetition (P3); the table is ordered before the
            // chain's first progress count, which every worker awaits before it reads it.
            if (p.grad) {
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
            const cell_t a = r16_chain<true>(p, sm, p.T, Tb, L, p.SP);
            stamp(p, 11);
            // nll = -log alpha[T_b-1, L_b-1] (NoBlankCTC.py:58-68,139)
            const float am = a.x;
            const float nll = am > 0.f ? -(__builtin_amdgcn_logf(am) + (float)(cell_k(a) - kXrBias)) * kLn2 : -kNeg;
            publish_and_reduce(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter, [](float x, int) { return x; });
            if (p.stop == -50 && lane == 0) {                // diagnostic: when the alpha wave (loss ticket) is done
                const int bid = blockIdx.x, nb = gridDim.x;
                const int slot = bid == 0 ? 6 : bid == nb / 2 ? 7 : bid == nb - 1 ? 8 : -1;
                if (slot >= 0) {
                    unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * slot;
                    o[0] = __builtin_amdgcn_s_memtime();
                    o[1] = __builtin_amdgcn_s_memrealtime();
                }
            }
        } else if (p.grad) {
            r16_chain<false>(p, sm, p.T, Tb, L, p.SP);
            stamp(p, 11);
        }
        return;
    }

    // ---------------------------------------------------------------- workers
    float *tile = sm.stage + (size_t)u * 4 * RP;             // this worker's staging / occupancy tile
    float *tile_row = tile + rho * RP + c_lane;              // this lane's pair j lives at tile_row + 32 j
    // states served by this lane in the two passes: l = i16 and l = 16 + i16 (S <= 31)
    int lst[2];
    float *gat[2];                                           // tile address of class lab[l] in this lane's row
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        lst[s] = 16 * s + i16;
        gat[s] = tile + rho * RP + (lst[s] < p.SP ? sm.lab[lst[s]] : 0);
    }
    const bool own[2] = {lst[0] < L, lst[1] < L};
    const float maskv = col_ok ? 0.f : ninf;
    cell_t *const spare_w = reinterpret_cast<cell_t *>(sm.dummy);
    float rs[G];                                             // grad_scale / sum_c exp(x - max), 0 for dead rows
#pragma unroll
    for (int g = 0; g < G; ++g) {                            // P1: extremes first
        f2_t *x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tb;
        float m = fmaxf(x[CH2 - 1].x, x[CH2 - 1].y) + maskv;
#pragma unroll
        for (int j = 0; j < CH2 - 1; ++j) m = fmaxf(m, fmaxf(x[j].x, x[j].y));
        row16_allmax(m);
        // raw rows -> tile, labels' logits back
#pragma unroll
        for (int j = 0; j < CH2; ++j) *reinterpret_cast<f2_t *>(tile_row + 32 * j) = x[j];
        lds_order();
        float xv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) xv[s] = *gat[s];
        lds_order();
        // the row registers become exp(x - max): P3 needs softmax(x) = that times 1/sum
        const float mb = -m * kLog2e;
        x[CH2 - 1].x = __builtin_amdgcn_exp2f(__builtin_fmaf(x[CH2 - 1].x + maskv, kLog2e, mb));
        x[CH2 - 1].y = __builtin_amdgcn_exp2f(__builtin_fmaf(x[CH2 - 1].y + maskv, kLog2e, mb));
        float sum = x[CH2 - 1].x + x[CH2 - 1].y;
#pragma unroll
        for (int j = 0; j < CH2 - 1; ++j) {
            x[j].x = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j].x, kLog2e, mb));
            x[j].y = __builtin_amdgcn_exp2f(__builtin_fmaf(x[j].y, kLog2e, mb));
            sum += x[j].x + x[j].y;
        }
        row16_allsum(sum);
        const float l2sum = __builtin_amdgcn_logf(sum);
        rs[g] = live ? p.grad_scale * __builtin_amdgcn_rcpf(sum) : 0.f;
        // emissions e = log_softmax(x)[lab_l] in log2 units, split into 2^floor * 2^frac
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float e2 = fmaxf(__builtin_fmaf(xv[s] - m, kLog2e, -l2sum), kXrMinLog2);
            const float fl = __builtin_floorf(e2);
            const float pm = __builtin_amdgcn_exp2f(e2 - fl);
            cell_t *dst = (live && lst[s] < p.SP) ? sm.em + lst[s] * sm.TP + t : spare_w;
            *dst = own[s] ? make_cell(pm, (int)fl) : zero;
        }
        lds_order();
        sm.cnt[u] = 4 * (g + 1);                             // publishes the four slots (same wave: in order)
    }
    stamp(p, 2);
    if (!p.grad) return;

    const int Tlive = Tb;
    const float gsc = p.grad_scale;
    const cell_t *const zero_r = sm.em + (size_t)p.SP * sm.TP;   // the spare state row of em stays zero
    const f2_t zero2 = {0.f, 0.f};

    // P3: middle-out, one look at the chains' progress per group
#pragma unroll
    for (int g = G - 1; g >= 0; --g) {
        int need_a = 0, need_b = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tk = pipe_row(p.T, u, 4 * g + k);      // scalar twin of tv[g]
            if (tk >= 0 && tk < Tlive) {
                need_a = max(need_a, tk + 1);
                need_b = max(need_b, Tlive - tk);
            }
        }
        if (need_a > 0) {
            int spins = 0;
            while ((*(lds_cvint *)(sm.cnt + kPipeWorkers) < need_a || *(lds_cvint *)(sm.cnt + kPipeWorkers + 1) < need_b) &&
                   ++spins < kSpinLimit)
                __builtin_amdgcn_s_sleep(8);
            lds_order();
        }
        if (p.stop < 0) stamp(p, 3 + (G - 1 - g));
        int occn[2] = {0, 0}, max_occ = 0;
        if (need_a > 0) {                                    // (wave-uniform; a group without live rows adds nothing)
            occn[0] = own[0] ? sm.occ[lst[0]] : 0;
            occn[1] = own[1] ? sm.occ[lst[1]] : 0;
            max_occ = __builtin_amdgcn_readfirstlane(sm.occ[(p.SP + 3) & ~3]);
        }
        f2_t *x = v[g];
        const int t = tv[g];
        const bool live = t >= 0 && t < Tlive;
        // gamma_t(l) = alpha_t(l) beta_t(l) / sum_l' (...): mantissa products, exponents added and
        // shifted by the row's largest
        float pr[2];
        int ks[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool in = live && lst[s] < p.SP;
            const int off = lst[s] * sm.TP + t;
            const cell_t a = *(in ? sm.al + off : zero_r), bb = *(in ? sm.be + off : zero_r);
            pr[s] = a.x * bb.x;
            ks[s] = cell_k(a) + cell_k(bb);
        }
        int km = max(pr[0] > 0.f ? ks[0] : 0, pr[1] > 0.f ? ks[1] : 0);
        row16_allmax(km);
        float z[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) z[s] = __builtin_amdgcn_ldexpf(pr[s], ks[s] - km);
        float tot = z[0] + z[1];
        row16_allsum(tot);
        const float rinv = (live && tot > 0.f) ? gsc * __builtin_amdgcn_rcpf(tot) : 0.f;
        // class occupancy of the four rows: zero the tile, scatter the scaled posteriors
#pragma unroll
        for (int j = 0; j < CH2; ++j) *reinterpret_cast<f2_t *>(tile_row + 32 * j) = zero2;
        lds_order();
        // (only lanes that own a state write; pass k adds the k-th repetition of a label)
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
        // dense rows: grad = softmax(x) * scale - occupancy   (dead rows: scale = occupancy = 0)
        if (t >= 0) {
            float *gp = p.grad + ((int64_t)t * p.B + b) * p.C + c_lane;
#pragma unroll
            for (int j = 0; j < CH2; ++j) {
                const f2_t occ = *reinterpret_cast<const f2_t *>(tile_row + 32 * j);
                f2_t gv;
                gv.x = __builtin_fmaf(x[j].x, rs[g], -occ.x);
                gv.y = __builtin_fmaf(x[j].y, rs[g], -occ.y);
                if (j < CH2 - 1 || col_ok) __builtin_nontemporal_store(gv, reinterpret_cast<f2_t *>(gp + 32 * j));
            }
        }
        lds_order();
    }
    stamp(p, 7);
    spread(1);
}

// ---- diagnostic probe (tools/chain_probe.py): the chains alone, every row already published ------
__global__ __launch_bounds__(kThreads, 4) void r16_chain_probe_kernel(NoblankParams p, unsigned long long *out, int waves_alive)
{
    extern __shared__ float4 smem_raw[];
    const R16Smem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, 32);
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    for (int i = tid; i < 3 * (p.SP + 1) * sm.TP; i += kThreads) sm.em[i - kR16Pad] = make_cell(1.5f, -1);
    if (tid < 16) sm.cnt[tid] = kPipeRows;                   // everything published
    __syncthreads();
    if (w >= waves_alive) return;
    if (w > 1) {                                             // bystanders: poll like a waiting worker
        typedef const volatile __attribute__((address_space(3))) int lds_cvint;
        int spins = 0;
        while (*(lds_cvint *)(sm.cnt + kPipeWorkers) < p.T && ++spins < 100000) __builtin_amdgcn_s_sleep(8);
        return;
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const cell_t c = w == 0 ? r16_chain<true>(p, sm, p.T, p.T, p.SP, p.SP) : r16_chain<false>(p, sm, p.T, p.T, p.SP, p.SP);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) {
        out[w] = t1 - t0;
        out[4 + w] = (unsigned long long)cell_k(c);
    }
}

}  // namespace ctc
