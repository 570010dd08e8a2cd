// binary_flow_kernel: the binary loss + gradient with the emissions STREAMED to the scans (round 3).
// Included by binary.hip (same arithmetic and helpers as binary_pipe_kernel; what changes is who does what, when, and on
// which unit).
//
// binary_pipe_kernel runs its phases one after another for the whole sample: logs of all rows, a workgroup barrier, all
// emission tiles, and only then do the scans start (11.5 us after entry at config 3, of 27.7); its emission tiles are fp32
// MFMA, which takes three quarters of the SIMD's issue time away from the VALU work of the waves beside it
// (tools/micro/mfma_rate.hip).  Here:
//
//   waves 0 / 1    the alpha / beta' scans: one tile flag per 16 steps; the lattice is kept in units of log2.
//   waves 2..5     TILE waves, one per SIMD, no rows of their own.  Emission tiles E = D . Y^T: (direction, parity) =
//                  (k & 1, k >> 1) makes the front tiles k>>1, k>>1 + 2, ... or the back tiles MT-1 - (k>>1), ... up to
//                  the middle of the sequence; a tile job starts when the owners of its 16 rows have published them.
//   waves 6..15    10 workers.  Worker u owns sixteen rows: slot (g, side), g = 0..7, is the front row 10g+u or the back
//                  row T-1-(10g+u).  Logs outside-in (the rows the scans need first), `done[u] = rounds finished` after
//                  each pair, no barrier, priority falling as the wave advances (a SIMD serves its oldest wave first:
//                  the youngest worker would otherwise hold every tile back).  Then the gradient, four rows at a time,
//                  middle-out as both scans pass them: posteriors, G = gamma . Y on v_mfma_f32_4x4x1, elementwise, stores.
//
// Multi-hot targets are exact in bf16, so their emission tiles run on v_mfma_f32_16x16x32_bf16 with the D row split into
// three bf16 terms on the fly (24 significant bits = 3 x 8: the split is exact, products of two bf16 are exact in fp32,
// accumulation is fp32).  Whether a sample's targets ARE exact in bf16 is found while they are staged; soft targets that
// are not take fp32 tiles (16x16x4, from the fp32 image) -- same kernel, same hand-offs, a uniform branch per workgroup.
// Hand-offs: single-writer LDS words only (done[u], tile[m], prog[2], ystage[w]); a wave's LDS operations complete in
// order, so a counter store after the row stores publishes them (common.hpp: lds_order).
#pragma once

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef CTC_FLOW_TILE_WAVES
#define CTC_FLOW_TILE_WAVES 4
#endif
constexpr int kFlowTileWaves = CTC_FLOW_TILE_WAVES;              // (even: half of them per direction)
constexpr int kFlowFirstWorker = 2 + kFlowTileWaves;
constexpr int kFlowWorkers = kBinWaves - kFlowFirstWorker;   // 10
constexpr int kFlowRounds = (80 + kFlowWorkers - 1) / kFlowWorkers;   // T <= 160
static_assert(kFlowRounds % 2 == 0, "two rounds per gradient group");
constexpr int kFlowGroups = kFlowRounds / 2;
constexpr int kFlowMaxT = 2 * kFlowWorkers * kFlowRounds;    // 160
constexpr int kFlowMaxTiles = kFlowMaxT / 16;                // 10
constexpr int kFlowFlags = 64;
constexpr int kFlowYRows = (65 + kFlowFirstWorker - 1) / kFlowFirstWorker;   // target rows one staging wave holds (S + 1 <= 65)

// pitches (in bf16 elements) of the two bf16 target images: Y16[label][class] (B operand of the emission tiles; one
// extra all-zero row that out-of-range labels read) and YT[class][label] (B operand of G).  Both = 8 (mod 32) elements
// = 4 (mod 16) words: the 16-byte fragment reads of sixteen lanes fall on sixteen different groups of four banks.
__host__ __device__ inline int flow_yp(int C) { return (C + 31) / 32 * 32 + 8; }
__host__ __device__ inline size_t flow_y_floats(int SP, int C, int PD)
{   // the fp32 image [SP][PD] (gradient contraction; emission tiles of soft targets), then the bf16 image Y16
    const size_t halfs = (size_t)(SP + 1) * flow_yp(C);
    return (((size_t)SP * PD + 3) & ~(size_t)3) + (((halfs + 1) / 2 + 3) & ~(size_t)3);
}

struct BinaryFlowSmem {
    float *em, *al, *be, *dummy, *q, *zrow, *ys, *dimg;
    unsigned short *y16;
    int *prog, *fail, *tile, *done, *ystage;
    __device__ BinaryFlowSmem(float *base, int T, int SP, int PD, int C)
    {
        em = base + kPrefetch * SP;
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        dummy = be + (size_t)T * SP;                         // [0..3] idle-lane slots, then the flag words (all zeroed at entry)
        int *f = reinterpret_cast<int *>(dummy);
        prog = f + 4;                                        // [2] completed steps of the alpha / beta' scans
        fail = f + 6;                                        // a bounded wait ran out somewhere: outputs are NaN
        tile = f + 8;                                        // [kFlowMaxTiles] emission tile m is in LDS
        done = f + 20;                                       // [kFlowWorkers] rounds of logs worker u has published
        ystage = f + 32;                                     // [kFlowFirstWorker] that wave's share of the bf16 target images is in LDS (1: exact, 2: not)
        q = dummy + kFlowFlags;
        zrow = q + ((T + 7) & ~3);                            // (q[T]: where idle slots put their sum)
        ys = zrow + 64;
        y16 = reinterpret_cast<unsigned short *>(ys + (((size_t)SP * PD + 3) & ~(size_t)3));
        dimg = ys + flow_y_floats(SP, C, PD);
    }
};

static size_t binary_flow_smem_bytes(int T, int SP, int PD, int C)
{
    return ((size_t)(3 * T + 2 * kPrefetch) * SP + kFlowFlags + ((T + 7) & ~3) + 64 + flow_y_floats(SP, C, PD) + (size_t)(T + 1) * PD) * 4;
}

// pitch of the D image (and of the fp32 target image): >= the K extent padded to 32, and = 4 (mod 32) -- the 16-byte
// fragment reads D[16m + (lane & 15)][.. + 4 or 8 (lane >> 4) ..] of sixteen lanes then fall on sixteen different groups
// of four banks
static int binary_flow_pitch(int C)
{
    int PD = (C + 15) / 16 * 16 + 4;
    while (PD % 32 != 4) PD += 4;
    return PD;
}

// f = t0 + t1 + t2 exactly, each term a bf16 (8 significant bits); eight floats -> three packed operands of the bf16 MFMA
__device__ __forceinline__ void split3_bf16(const float (&f)[8], u32x4 &t0, u32x4 &t1, u32x4 &t2)
{
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float a = f[2 * h], b = f[2 * h + 1];
        const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
        t0[h] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);  // (high halves of b and a: truncation to bf16)
        const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
        const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
        t1[h] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u), sb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
        t2[h] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4 &a, const u32x4 &b, const f32x4 &c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CH, bool WT, bool GAMMA = false>
__global__ __launch_bounds__(kBinThreads) void binary_flow_kernel(BinaryParams p, int PD)
{
    extern __shared__ float4 smem_raw[];
    const BinaryFlowSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, PD, p.C);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64) note_arrival(p.counter, b);                  // (wave 1: the beta' scan)
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    stamp(p, 0);
    const int H = (p.T + 1) >> 1;                            // rows [0,H): front rows, [H,T): back rows
    const int NG = (H + kFlowWorkers - 1) / kFlowWorkers;    // rounds in use
    const int NGR = (NG + 1) >> 1;                           // gradient groups (two rounds each) in use
    const int MT = (p.T + 15) >> 4;                          // tiles of 16 rows
    const int u = w - kFlowFirstWorker;                      // worker index
    auto slot_row_of = [&](int uu, int g, int side) {        // the row of worker uu's slot (g, side), or -1
        const int i = kFlowWorkers * g + uu;
        const int t = side ? p.T - 1 - i : i;
        return (side ? t >= H : t < H) ? t : -1;
    };
    auto slot_row = [&](int g, int side) { return slot_row_of(u, g, side); };
    // The outermost group (rounds 0 and 1: the rows both scans reach LAST) of four workers of SIMDs 2 and 3 is finished by the two
    // tile waves of SIMDs 0 and 1, which have been idle since their last tile and whose SIMDs are free once the scans are done:
    // ten gradient groups per SIMD instead of twelve / eight.  (helper wave 4 <- workers 0, 1; wave 5 <- workers 4, 5)
#ifdef CTC_X_FLOW_NO_DELEGATE
    constexpr bool kDelegate = false;
#else
    constexpr bool kDelegate = !GAMMA && kFlowTileWaves == 4;
#endif
#ifdef CTC_X_FLOW_DELEGATE_OWN                               // (measured: the tile waves of SIMDs 2 and 3 taking a group of their OWN SIMD's
    constexpr bool kDelegateOwn = true;                      // third worker as well -- no capacity gained, 22.3 against 22.15 us)
#else
    constexpr bool kDelegateOwn = false;
#endif
    const bool delegated = kDelegate && p.grad && (u == 0 || u == 1 || u == 4 || u == 5 || (kDelegateOwn && (u == 8 || u == 9)));

    // rows of this worker, issued before anything else (resident until the gradient: x, then 1 + exp(-x)).
    // Every load is consumed on every path that issued it (binary_pipe_kernel: the vmcnt trap).
    float v[kFlowRounds][2][CH];
    if (w >= kFlowFirstWorker) {
#pragma unroll
        for (int g = 0; g < kFlowRounds; ++g) {
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
    // y[b], held in registers by the six waves without rows (wave w: target rows w, w + 6, ...): the loads go out with
    // the workers' row loads, before anything waits
    const int YP = flow_yp(p.C);
    float yv[kFlowYRows][CH];
    if (w < kFlowFirstWorker) {
        const float *yb = p.y + (int64_t)b * p.S * p.C;
#pragma unroll
        for (int i = 0; i < kFlowYRows; ++i) {
            const int l = w + kFlowFirstWorker * i;
            if (l <= p.SP) {                                 // (uniform; row SP: the zero row of the bf16 image)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    yv[i][j] = (l < p.S && c < p.C) ? yb[l * p.C + c] : 0.f;
                }
            }
        }
    }
    if (tid < kFlowFlags) sm.dummy[tid] = 0.f;               // (every flag word)
    if (tid >= 128 && tid < 192) sm.zrow[tid - 128] = 0.f;
    for (int i = tid; i < kPrefetch * p.SP; i += kBinThreads) {
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }
    __syncthreads();                                         // the flags exist (no wave has waited for memory yet)
    const ScalarLengths lens(p.in_len + b, p.tgt_len + b);

    // Targets -> LDS: the fp32 image [SP][PD] (the gradient's contraction reads it) and the bf16 image Y16[SP + 1][YP] of
    // their high halves (B operand of the emission tiles; row SP is zeros), zero beyond S labels / C classes (MFMA K and N
    // padding).  Each wave publishes with its share whether its values were exact in bf16 (ystage[w] = 1) or not (2).
    auto six = [&](const int *f, int equals) -> bool {       // does one of the six words equal `equals`?
        return __builtin_amdgcn_ballot_w64(lane < kFlowFirstWorker && *(lds_cvint *)(f + (lane < kFlowFirstWorker ? lane : 0)) == equals) != 0;
    };
    bool exact = true;
    auto wait_y = [&]() -> bool {                            // -> false: the wait ran out; sets `exact`
        int spins = 0;
        while (six(sm.ystage, 0)) {
            if (++spins >= (1 << 20)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        exact = !six(sm.ystage, 2);
        lds_order();
        return true;
    };
    if (w < kFlowFirstWorker) {
        bool inexact = false;
#pragma unroll
        for (int i = 0; i < kFlowYRows; ++i) {
            const int l = w + kFlowFirstWorker * i;
            if (l <= p.SP) {                                 // (uniform)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    const unsigned bits = __builtin_bit_cast(unsigned, yv[i][j]);
                    inexact = inexact || (bits & 0xffffu) != 0;
                    if (c < YP) sm.y16[l * YP + c] = (unsigned short)(bits >> 16);
                    if (l < p.SP && c < PD) sm.ys[l * PD + c] = yv[i][j];
                }
            }
        }
        lds_order();
        sm.ystage[w] = __builtin_amdgcn_ballot_w64(inexact) == 0 ? 1 : 2;
    }
    int64_t Tb64, L64;
    lens.get(Tb64, L64);
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;
    const float invC = 1.0f / (float)p.C;
    const float inv2 = invC * kLog2e;                        // emissions, alpha and beta' are kept in units of log2
    const float gs = p.grad_scale * invC;
    stamp(p, 1);

    // ---------------- the gradient of a group of four rows (the workers; two tile waves for eight delegated groups) ----------------
    bool starved = false;
    bool have_lse = false;
    float c2 = 0.f;                                          // -(log2 of the rows' normaliser)
    const unsigned voff = 4u * lane;
    const int lane_l = lane < p.SP ? lane : 0;
    const bool in_l = lane < L;
    const float ninf = -__builtin_inff();
    // (a worker's OWN groups: the rows of every group are made once, before its gradient loop, and kept as opaque scalars --
    // the chain of compares and selects below is ~45 scalar instructions, a wave issues one instruction at a time whatever
    // its kind, and the gradient half is bound by issue; `own` selects the cached values)
    int c_tt[kFlowGroups][4], c_tl[kFlowGroups][4], c_hi[kFlowGroups], c_lo[kFlowGroups];
    int c_ro[kFlowGroups][4];                                // max(tl, 0) * SP: the lattice row of a live slot (row 0 for an idle one)
    auto row_off = [&](int jg, int i, int tl_i, bool own) { return own ? c_ro[jg][i] : (tl_i >= 0 ? tl_i : 0) * p.SP; };
    auto rows_of = [&](int uu, int jg, int (&tt)[4], int (&tl)[4], int &t_hi, int &t_lo, bool own = false) {
        if (own) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { tt[i] = c_tt[jg][i]; tl[i] = c_tl[jg][i]; }
            t_hi = c_hi[jg]; t_lo = c_lo[jg];
            return;
        }
        t_hi = -1; t_lo = p.T;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tt[i] = slot_row_of(uu, 2 * jg + (i & 1), i >> 1);
            tl[i] = tt[i] < Tb ? tt[i] : -1;
            if (tl[i] >= 0) {
                t_hi = tl[i] > t_hi ? tl[i] : t_hi;
                t_lo = tl[i] < t_lo ? tl[i] : t_lo;
            }
        }
    };
    auto post_ready = [&](int uu, int jg, bool own = false) -> bool {   // have both scans passed the rows of the group?
        int tt[4], tl[4], t_hi, t_lo;
        rows_of(uu, jg, tt, tl, t_hi, t_lo, own);
        if (t_hi < 0) return true;
        const int pa = *(lds_cvint *)sm.prog, pb = *(lds_cvint *)(sm.prog + 1), bad = *(lds_cvint *)sm.fail;
        if (bad) starved = true;
        return bad || (pa >= t_hi + 1 && pb >= Tb - t_lo);
    };
    auto post = [&](int uu, int jg, bool own = false) {
        int tt[4], tl[4], t_hi, t_lo;
        rows_of(uu, jg, tt, tl, t_hi, t_lo, own);
        if (t_hi < 0) return;                                // (uniform) no live row
        lds_order();
        if (p.SP <= 32) {                                    // (uniform) two rows per register, side by side in the halves of the wave:
            const int hl = lane & 31;                        // half as many loads, additions, exponentials and reduction steps
            const bool up = lane >= 32;
            const bool in_h = hl < L;
            int tr[2];
            float z[2], pe[2], sum[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                tr[r] = up ? tl[2 * r + 1] : tl[2 * r];
                const int off = (up ? row_off(jg, 2 * r + 1, tl[2 * r + 1], own) : row_off(jg, 2 * r, tl[2 * r], own)) + (hl < p.SP ? hl : 0);
                const float za = sm.al[off], zb = sm.be[off], ze = sm.em[off];
                z[r] = (in_h && tr[r] >= 0) ? za + zb - ze : ninf;
            }
            if (!have_lse) {                                 // (uniform) first group of this worker
                float m[2] = {z[0], z[1]};
                halves_max2(m[0], m[1], up);
#pragma unroll
                for (int r = 0; r < 2; ++r) pe[r] = __builtin_amdgcn_exp2f(z[r] - (tr[r] >= 0 ? m[r] : 0.f));
#pragma unroll
                for (int r = 0; r < 2; ++r) sum[r] = pe[r];
                halves_sum2(sum[0], sum[1], up);
#pragma unroll
                for (int k = 3; k >= 0; --k)                 // (every live row's normaliser is the same number: take one)
                    if (tl[k] >= 0) {                        // (uniform)
                        const float lse = -m[k >> 1] - __builtin_amdgcn_logf(sum[k >> 1]);
                        c2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lse), (k & 1) ? 32 : 0));
                    }
                have_lse = true;
            } else {
#pragma unroll
                for (int r = 0; r < 2; ++r) pe[r] = __builtin_amdgcn_exp2f(z[r] + c2);
#pragma unroll
                for (int r = 0; r < 2; ++r) sum[r] = pe[r];
                halves_sum2(sum[0], sum[1], up);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float inv = __builtin_amdgcn_rcpf(sum[r]);
                if (hl < p.SP && tr[r] >= 0)
                    sm.be[(up ? row_off(jg, 2 * r + 1, tl[2 * r + 1], own) : row_off(jg, 2 * r, tl[2 * r], own)) + hl] = pe[r] * (gs * inv);
                if (GAMMA && hl < p.S && tr[r] >= 0)         // posteriors output: gamma_t(l), rows sum to 1
                    p.gamma[((int64_t)b * p.T + tr[r]) * p.S + hl] = starved ? __builtin_nanf("") : pe[r] * inv;
            }
            return;
        }
        float z[4], pe[4], sum[4];
        {   // (idle slots read row 0 and are masked: twelve loads in flight, one wait)
            float za[4], zb[4], ze[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int off = row_off(jg, k, tl[k], own) + lane_l;
                za[k] = sm.al[off]; zb[k] = sm.be[off]; ze[k] = sm.em[off];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) z[k] = (in_l && tl[k] >= 0) ? za[k] + zb[k] - ze[k] : ninf;
        }
        if (!have_lse) {                                     // (uniform) first group of this worker
            float m[4] = {z[0], z[1], z[2], z[3]};
            wave_max4(m[0], m[1], m[2], m[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) pe[k] = __builtin_amdgcn_exp2f(z[k] - (tl[k] >= 0 ? m[k] : 0.f));
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] = pe[k];
            wave_sum4(sum[0], sum[1], sum[2], sum[3]);
#pragma unroll
            for (int k = 3; k >= 0; --k)
                if (tl[k] >= 0) c2 = -m[k] - __builtin_amdgcn_logf(sum[k]);   // (uniform)
            have_lse = true;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) pe[k] = __builtin_amdgcn_exp2f(z[k] + c2);
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] = pe[k];
            wave_sum4(sum[0], sum[1], sum[2], sum[3]);
        }
        if (lane < p.SP) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tl[k] >= 0) sm.be[row_off(jg, k, tl[k], own) + lane] = pe[k] * (gs * __builtin_amdgcn_rcpf(sum[k]));
        }
        if (GAMMA && lane < p.S) {                           // posteriors output: gamma_t(l), rows sum to 1
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tl[k] >= 0)
                    p.gamma[((int64_t)b * p.T + tl[k]) * p.S + lane] = starved ? __builtin_nanf("") : pe[k] * __builtin_amdgcn_rcpf(sum[k]);
        }
    };
    auto elem = [&](int uu, int jg, unsigned slow4, auto &&P, bool own = false) {   // slow4: the careful-path bits of the group's four slots; P(r, side, j): their sigmoids
        int tt[4], tl[4], t_hi, t_lo;
        rows_of(uu, jg, tt, tl, t_hi, t_lo, own);
        if (starved) raise_status(p.counter, kStatusBinaryStarved);
        if constexpr (GAMMA) {                               // posteriors only: rows beyond T_b get zeros, no gradient
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (tt[i] >= 0 && tl[i] < 0 && lane < p.S) p.gamma[((int64_t)b * p.T + tt[i]) * p.S + lane] = 0.f;
            return;
        }
        lds_order();
        f32x4 acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t_hi >= 0) {
            {                                                // G = gamma . Y: four labels per batch, two batches per trip
                const int ti = tl[lane & 3];
                const float *arow = ti >= 0 ? sm.be + ti * p.SP : sm.zrow;   // (an idle slot contributes a row of zeros)
                const float *yrow = sm.ys + lane;
                auto operands = [&](int l0, float4 &fa, float (&fy)[4][CH]) {
                    fa = *reinterpret_cast<const float4 *>(arow + (l0 < p.SP ? l0 : 0));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < CH; ++j) fy[i][j] = yrow[(l0 + i < p.SP ? l0 + i : p.SP - 1) * PD + 64 * j];
                };
                auto contract = [&](const float4 &fa, const float (&fy)[4][CH]) {
                    const float fav[4] = {fa.x, fa.y, fa.z, fa.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < CH; ++j) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(fav[i], fy[i][j], acc[j], 0, 0, 0);
                };
                float4 fa0, fa1;
                float fy0[4][CH], fy1[4][CH];
                operands(0, fa0, fy0);
                for (int l0 = 0; l0 < L; l0 += 8) {
                    operands(l0 + 4, fa1, fy1);
                    contract(fa0, fy0);
                    if (l0 + 4 < L) {                        // (uniform)
                        operands(l0 + 8, fa0, fy0);
                        contract(fa1, fy1);
                    }
                }
            }
        }
        // the usual group -- four live rows, none with tails -- in one piece: one uniform branch instead of a dozen
        if (!starved && slow4 == 0 && tl[0] >= 0 && tl[1] >= 0 && tl[2] >= 0 && tl[3] >= 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *g = p.grad + ((int64_t)tl[i] * p.B + b) * p.C;
                bin_row_base_settled(g);
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float pr = P(i & 1, i >> 1, j);
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, __builtin_fmaf(pr, gs, -acc[j][i]));
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tt[i] < 0) continue;                         // wave-uniform
            float *g = p.grad + ((int64_t)tt[i] * p.B + b) * p.C;
            bin_row_base_settled(g);
            const int slot = 2 * (i & 1) + (i >> 1);         // (bit of slow4: slots (2jg, 0), (2jg, 1), (2jg+1, 0), (2jg+1, 1))
            if (tl[i] < 0 || starved) {                      // (uniform) dead row: zeros; starved: NaN
                const float fill = starved ? __builtin_nanf("") : 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j)
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, fill);
            } else if ((slow4 >> slot) & 1) {                 // (uniform) a row with tails: torch's floored denominator
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float pr = P(i & 1, i >> 1, j);
                    const float pq = pr * (1.0f - pr);
                    const float gv = __builtin_fmaf(pr, gs, -acc[j][i]) * (pq * __builtin_amdgcn_rcpf(fmaxf(pq, 1e-12f)));
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, gv);
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float pr = P(i & 1, i >> 1, j);
                    if (j < CH - 1 || lane + 64 * j < p.C) bin_store_col<WT>(g, voff, j, __builtin_fmaf(pr, gs, -acc[j][i]));
                }
            }
        }
    };

    // ---------------- the two scans ----------------
    if (w < 2) {
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
                if (rot) lattice_chain_fed<true, true, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, sm.prog, ready);
                else lattice_chain_fed<true, false, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP, sm.prog, ready);
            } else {
                if (rot) lattice_chain_fed<false, true, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, sm.prog + 1, ready);
                else lattice_chain_fed<false, false, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP, sm.prog + 1, ready);
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
            float nll = ok ? -kLn2 * sm.al[(Tb - 1) * p.SP + (L - 1)] : -kNeg;   // (the lattice is in units of log2)
            if (starved || *(lds_cvint *)sm.fail) {          // a bounded wait ran out: NaN, not a plausible number
                nll = __builtin_nanf("");
                raise_status(p.counter, kStatusBinaryStarved);
            }
            publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        }
        stamp(p, 7);
        return;
    }

    // ---------------- the tile waves ----------------
    if (w < kFlowFirstWorker) {
        __builtin_amdgcn_s_setprio(2);
        const int k = w - 2, back = k & 1, par = k >> 1;
        const bool helper = kDelegate && p.grad && (k >= 2 || kDelegateOwn);
        const int c0 = k == 2 ? 0 : k == 3 ? 4 : k == 0 ? 8 : 9;   // its clients: workers c0 (and c0 + 1 for waves 4 and 5)
        const int ncl = k >= 2 ? 2 : 1;
        float xh[2][2][2][CH];                               // [client][round 0 / 1][side][chunk]: logits, then sigmoids
        if (helper) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < ncl)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        const int t = slot_row_of(c0 + c, r, side);
                        const float *row = p.x + (int64_t)(t >= 0 ? t : 0) * p.st + (int64_t)b * p.sb;
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            const int cc = lane + 64 * j;
                            xh[c][r][side][j] = row[cc < p.C ? cc : p.C - 1];
                        }
                    }
        }
        const int MF = (MT + 1) >> 1;                        // tiles [0, MF) are made front to back, [MF, MT) back to front
        const int NT = (p.SP + 15) >> 4;
        const int fr = lane & 15, fq = lane >> 4;
        bool bad = !wait_y();
#ifdef CTC_AMD_FAULT_INJECT                                  // tests/test_status.py: sample 2's first tile wave "starves"
        if (b == 2 && w == 2) bad = true;
#endif
        // E = D . Y^T, 16 rows x all labels per job
        int nt = 0;                                          // (diagnostics: which tile of this wave)
        for (int jt = par;; jt += kFlowTileWaves / 2, ++nt) {
            const int m = back ? MT - 1 - jt : jt;
            if (back ? m < MF : m >= MF) break;              // (uniform)
            {   // the sixteen rows of the tile: has their owner published the round they belong to?
                const int tr = 16 * m + fr;
                const int tc = tr < p.T ? tr : p.T - 1;
                const int i = tc < H ? tc : p.T - 1 - tc;
                const int g = i / kFlowWorkers, uu = i - g * kFlowWorkers;
                int spins = 0;
                while (__builtin_amdgcn_ballot_w64(*(lds_cvint *)(sm.done + uu) <= g) != 0) {
                    if (++spins >= (1 << 21)) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                lds_order();
            }
            stamp(p, nt == 0 ? 8 : nt == 1 ? 4 : 6);
            const int ta = 16 * m + fr;
            const float *arow = sm.dimg + (ta < p.T ? ta : p.T - 1) * PD;
            auto epilogue = [&](auto N, const f32x4 *acc) {    // branch-free: lanes outside the tile store into an idle slot
                constexpr int NN = decltype(N)::value;
                const float4 q4 = *reinterpret_cast<const float4 *>(sm.q + 16 * m + 4 * fq);
                const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const int lrow = 16 * n + fr;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int t = 16 * m + 4 * fq + i;
                        float *dst = (t < Tb && lrow < p.SP) ? sm.em + t * p.SP + lrow : sm.dummy + i;
                        *dst = lrow < L ? (acc[n][i] + qv[i]) * inv2 : kNeg;
                    }
                }
            };
            auto tile_bf16 = [&](auto N) {                   // targets exact in bf16: three bf16 terms of the D row
                constexpr int NN = decltype(N)::value;
                const int KS = (p.C + 31) >> 5;
                const float *ap = arow + 8 * fq;
                const unsigned short *bq[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const int lrow = 16 * n + fr;
                    bq[n] = sm.y16 + (lrow < p.SP ? lrow : p.SP) * YP + 8 * fq;   // (row SP: zeros)
                }
                f32x4 acc[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                float4 x0 = *reinterpret_cast<const float4 *>(ap), x1 = *reinterpret_cast<const float4 *>(ap + 4);
                u32x4 yb[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) yb[n] = *reinterpret_cast<const u32x4 *>(bq[n]);
                for (int ks = 0; ks < KS; ++ks) {
                    const float f[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    u32x4 yc[NN];
#pragma unroll
                    for (int n = 0; n < NN; ++n) yc[n] = yb[n];
                    const int kn = ks + 1 < KS ? ks + 1 : ks;            // (the read past the end repeats the last batch)
                    x0 = *reinterpret_cast<const float4 *>(ap + 32 * kn);
                    x1 = *reinterpret_cast<const float4 *>(ap + 32 * kn + 4);
#pragma unroll
                    for (int n = 0; n < NN; ++n) yb[n] = *reinterpret_cast<const u32x4 *>(bq[n] + 32 * kn);
                    __builtin_amdgcn_sched_barrier(0);       // (the next batch's reads stay in front of this batch's arithmetic)
                    u32x4 t0, t1, t2;
                    split3_bf16(f, t0, t1, t2);
#pragma unroll
                    for (int n = 0; n < NN; ++n) acc[n] = mfma_bf16(t0, yc[n], acc[n]);
#pragma unroll
                    for (int n = 0; n < NN; ++n) acc[n] = mfma_bf16(t1, yc[n], acc[n]);
#pragma unroll
                    for (int n = 0; n < NN; ++n) acc[n] = mfma_bf16(t2, yc[n], acc[n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                epilogue(N, acc);
            };
            auto tile_f32 = [&](auto N) {                    // soft targets: fp32 tiles from the fp32 image
                constexpr int NN = decltype(N)::value;
                const int KB = (p.C + 15) >> 4;
                const float *ap = arow + 4 * fq;
                const float *bp[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) {
                    const int lrow = 16 * n + fr;
                    bp[n] = sm.ys + (lrow < p.SP ? lrow : p.SP - 1) * PD + 4 * fq;
                }
                f32x4 acc[NN];
#pragma unroll
                for (int n = 0; n < NN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int kb = 0; kb < KB; ++kb) {
                    const float4 fa = *reinterpret_cast<const float4 *>(ap + 16 * kb);
                    const float a4[4] = {fa.x, fa.y, fa.z, fa.w};
                    float4 fb[NN];
#pragma unroll
                    for (int n = 0; n < NN; ++n) fb[n] = *reinterpret_cast<const float4 *>(bp[n] + 16 * kb);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int n = 0; n < NN; ++n) {
                            const float b4[4] = {fb[n].x, fb[n].y, fb[n].z, fb[n].w};
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], b4[i], acc[n], 0, 0, 0);
                        }
                }
                epilogue(N, acc);
            };
            using std::integral_constant;
            if (exact) {
                switch (NT) {
                    case 1: tile_bf16(integral_constant<int, 1>{}); break;
                    case 2: tile_bf16(integral_constant<int, 2>{}); break;
                    case 3: tile_bf16(integral_constant<int, 3>{}); break;
                    default: tile_bf16(integral_constant<int, 4>{}); break;
                }
            } else {
                switch (NT) {
                    case 1: tile_f32(integral_constant<int, 1>{}); break;
                    case 2: tile_f32(integral_constant<int, 2>{}); break;
                    case 3: tile_f32(integral_constant<int, 3>{}); break;
                    default: tile_f32(integral_constant<int, 4>{}); break;
                }
            }
            lds_order();
            if (bad) *sm.fail = 1;
            lds_order();
            sm.tile[m] = 1;                                  // (every lane, same value)
            stamp(p, nt == 0 ? 3 : nt == 1 ? 9 : 10);
        }
        if (helper) {                                        // the delegated groups: sigmoids now, the rest when the scans are done
            __builtin_amdgcn_s_setprio(0);
            starved = bad;
            unsigned slow4[2] = {0u, 0u};
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if (c < ncl)
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int side = 0; side < 2; ++side) {
                        float far = 0.f;
#pragma unroll
                        for (int j = 0; j < CH; ++j) far = fmaxf(far, fabsf(xh[c][r][side][j] + 5.0f));
                        const bool careful = __builtin_amdgcn_ballot_w64(!(far < 11.0f)) != 0;   // (the workers' own test)
                        if (careful) slow4[c] |= 1u << (2 * r + side);
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            float sv;
                            if (careful) {
                                float lp, lq;
                                bce_logs_fast_s(xh[c][r][side][j], sv, lp, lq);
                            } else {
                                sv = 1.0f + __builtin_amdgcn_exp2f(xh[c][r][side][j] * -kLog2e);
                            }
                            xh[c][r][side][j] = __builtin_amdgcn_rcpf(sv);
                        }
                    }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (c >= ncl) break;
                int spins = 0;
                while (!starved && !post_ready(c0 + c, 0)) {
                    if (++spins >= (1 << 20)) starved = true;
                    __builtin_amdgcn_s_sleep(8);
                }
                post(c0 + c, 0);
                elem(c0 + c, 0, slow4[c], [&](int r, int side, int j) { return xh[c][r][side][j]; });
            }
            stamp(p, 11);
        }
        return;
    }

    // ---------------- workers: the logs of the rows -> D image, Q; outside-in, published round by round ----------------
    // (arithmetic: see binary_pipe_kernel, P1a)
    unsigned slow = 0;                                       // bit 2g+side: that row took the careful path
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int g = 0; g < kFlowRounds; ++g) {
        if (g < NG) {                                        // (uniform)
            float ql[2];
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int t = slot_row(g, side);
                float *drow = sm.dimg + (t >= 0 ? t : p.T) * PD + lane;   // (an idle slot writes the spare row behind the image)
                float far = 0.f;                             // max_j |x_j + 5|: < 11 <=> every x_j in (-16, 6)
#pragma unroll
                for (int j = 0; j < CH; ++j) far = fmaxf(far, fabsf(v[g][side][j] + 5.0f));
                if (__builtin_amdgcn_ballot_w64(!(far < 11.0f)) != 0) slow |= 1u << (2 * g + side);
                if (!((slow >> (2 * g + side)) & 1)) {
                    float prod = 1.0f, sx = -0.f;                // (-0 + x is x for every x: the first addition folds away)
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int c = lane + 64 * j;
                        const float xv = v[g][side][j];
#ifdef CTC_X_FLOW_NOEXP                                      // (ablation: are the logs bound by their arithmetic or by the rows' arrival?)
                        const float sv = 1.0f + xv * xv;
#else
                        const float sv = 1.0f + __builtin_amdgcn_exp2f(xv * -kLog2e);
#endif
                        v[g][side][j] = sv;
                        const bool in = j < CH - 1 || c < p.C;          // (only the last chunk can pass C)
                        prod *= in ? sv : 1.0f;
                        sx += in ? xv : 0.f;
                        if (j < CH - 1 || c < PD) drow[64 * j] = in ? xv : 0.f;   // zero K padding
                    }
#ifdef CTC_X_FLOW_NOEXP
                    ql[side] = prod - sx;
#else
                    ql[side] = __builtin_fmaf(__builtin_amdgcn_logf(prod), -kLn2, -sx);
#endif
                } else {
                    ql[side] = 0.f;
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const int c = lane + 64 * j;
                        float sv, lp, lq;
                        bce_logs_fast_s(v[g][side][j], sv, lp, lq);
                        v[g][side][j] = sv;
                        const bool in = j < CH - 1 || c < p.C;
                        ql[side] += in ? lq : 0.f;
                        if (j < CH - 1 || c < PD) drow[64 * j] = in ? lp - lq : 0.f;
                    }
                }
            }
            wave_sum2(ql[0], ql[1]);                         // (both rows of the round in one pass of DPP steps)
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int t = slot_row(g, side);
                if (lane == 0) sm.q[t >= 0 ? t : p.T] = ql[side];
            }
            lds_order();
            sm.done[u] = g + 1;                              // both rows of the round are in LDS (same wave: in order)
            // a SIMD serves its oldest ready wave first: without this the youngest worker of each SIMD finishes its logs
            // 4 us after the others, and every tile has a row of it.  Waves that are behind get the higher priority.
            if (g == 1) __builtin_amdgcn_s_setprio(2);
            if (g == 3) __builtin_amdgcn_s_setprio(1);
            if (g == 5) __builtin_amdgcn_s_setprio(0);
            if (g == 1) stamp(p, 2);
        }
    }
    __builtin_amdgcn_s_setprio(0);
    stamp(p, 6);
    if (!p.grad && !GAMMA) return;


    // ---------------- workers: the gradient, four rows at a time (binary_pipe_kernel, P3) ----------------
    // Groups middle-out as both scans pass them.  Per group: the posteriors of the four rows (one label per lane), gamma *
    // scale written over the rows of `be`; G = gamma . Y on v_mfma_f32_4x4x1 (16 blocks of 4x4, K = 1: the output column
    // is the lane and the four rows are the four accumulator registers -- the layout of the resident rows); then per
    // element a reciprocal, one fma and the store.
    starved = !wait_y();
    // the sigmoids of the resident rows, now, while the scans are on their way to the middle of the sequence
#pragma unroll
    for (int g = 0; g < kFlowRounds; ++g)
        if (g < NG)
#pragma unroll
            for (int side = 0; side < 2; ++side)
#pragma unroll
                for (int j = 0; j < CH; ++j) v[g][side][j] = __builtin_amdgcn_rcpf(v[g][side][j]);
#pragma unroll
    for (int jg = 0; jg < kFlowGroups; ++jg) {
        rows_of(u, jg, c_tt[jg], c_tl[jg], c_hi[jg], c_lo[jg]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { c_tt[jg][i] = opaque_s(c_tt[jg][i]); c_tl[jg][i] = opaque_s(c_tl[jg][i]); }
        c_hi[jg] = opaque_s(c_hi[jg]); c_lo[jg] = opaque_s(c_lo[jg]);
#pragma unroll
        for (int i = 0; i < 4; ++i) c_ro[jg][i] = opaque_s((c_tl[jg][i] >= 0 ? c_tl[jg][i] : 0) * p.SP);
    }
    stamp(p, 3);
#pragma unroll
    for (int jg = kFlowGroups - 1; jg >= 0; --jg) {          // rows nearest the middle are ready first
        if (jg < NGR && !(jg == 0 && delegated)) {           // (uniform)
            int spins = 0;
            while (!starved && !post_ready(u, jg, true)) {
                if (++spins >= (1 << 20)) starved = true;
                __builtin_amdgcn_s_sleep(8);
            }
            stamp(p, jg == 3 ? 8 : jg == 2 ? 4 : jg == 1 ? 5 : 9);
#ifndef CTC_X_FLOW_NO_GPRIO
            if (jg >= 2) __builtin_amdgcn_s_setprio(2);      // (workers that are behind go first: 25.5 -> 24.9 us at config 3)
            else if (jg == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
#endif
            post(u, jg, true);
            elem(u, jg, (slow >> (4 * jg)) & 15u, [&](int r, int side, int j) { return v[2 * jg + r][side][j]; }, true);
        }
    }
    stamp(p, 7);
}

template <bool WT, bool GAMMA>
static int launch_binary_flow_wt(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int PD)
{
    const dim3 grid(p.B), block(kBinThreads);
    switch (ch) {
        case 1: return launch<binary_flow_kernel<1, WT, GAMMA>>(grid, block, smem, s, p, PD);
        case 2: return launch<binary_flow_kernel<2, WT, GAMMA>>(grid, block, smem, s, p, PD);
        case 3: return launch<binary_flow_kernel<3, WT, GAMMA>>(grid, block, smem, s, p, PD);
        default: return launch<binary_flow_kernel<4, WT, GAMMA>>(grid, block, smem, s, p, PD);
    }
}

static int launch_binary_flow(int ch, size_t smem, hipStream_t s, const BinaryParams &p, int PD)
{
    const bool wt = (size_t)8 * p.T * p.B * p.C <= ((size_t)230 << 20);   // logits + gradient within the memory-side cache
    return wt ? launch_binary_flow_wt<true, false>(ch, smem, s, p, PD) : launch_binary_flow_wt<false, false>(ch, smem, s, p, PD);
}
