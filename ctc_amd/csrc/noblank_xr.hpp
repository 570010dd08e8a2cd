// Extended-range LINEAR-domain variant of the pipelined no-blank kernel (included by noblank.hip).
//
// Same schedule as noblank_pipe.hpp (two chain waves fed by 14 worker waves through LDS
// counters), different arithmetic on the lattice.  The log-domain recursion
//     alpha_t(l) = LSE(alpha_{t-1}(l), alpha_{t-1}(l-1)) + e_t(l)         (NoBlankCTC.py:71-87)
// has eight dependent VALU operations per step, two of them transcendental (~110 cycles per
// step, the critical path of the whole kernel at B = #CUs).  Here every lattice cell is the
// SAME number kept as a pair (m, k) = m * 2^k, m an fp32 mantissa, k an int32 exponent:
//     (m, k)_t(l) = ( (m 2^k)_{t-1}(l) + (m 2^k)_{t-1}(l-1) ) * p_t(l),   p = exp(e) = pm * 2^pk
// i.e. one integer max, two v_ldexp_f32, one add, one multiply per step and a v_frexp
// renormalisation every fourth step -- no exp, no log, about 40 % of the latency.  Unlike the
// textbook "scaled forward-backward" (one scale per time step), the exponent is PER STATE, so
// states may differ by any factor (2^+-(2^29)) exactly as in the log domain: no underflow, no
// data-dependent fallback.  The posterior gamma_t(l) ~ alpha_t(l) beta_t(l) then needs one
// multiply and one v_ldexp per cell (no exp), and nll = -(log2 m + k) ln 2 at (T_b-1, L_b-1).
//
// Exponents are stored biased by kXrBias so that "no mass" (m = 0) carries exponent 0 = minus
// infinity: a DPP shift with bound_ctrl (out-of-range lanes read 0) then supplies the correct
// neighbour for the first / last state for free, and a zero state can never win the max.
// Emissions below 2^-(2^21) (e < -1.45e6 nats) are clamped there so that T * pk cannot wrap.
//
// LDS cells are 8 bytes, so this variant needs (3T + 8) * SP * 8 bytes; shapes beyond that stay
// on the log-domain kernel (noblank_pipe.hpp).
#pragma once

namespace ctc {

typedef float cell_t __attribute__((ext_vector_type(2)));    // .x mantissa, .y exponent (int bits)
constexpr int kXrBias = 1 << 29;
constexpr float kXrMinLog2 = -2097152.0f;                   // -2^21

__device__ __forceinline__ cell_t make_cell(float m, int k)
{
    cell_t c;
    c.x = m;
    c.y = __builtin_bit_cast(float, k);
    return c;
}
__device__ __forceinline__ int cell_k(cell_t c)
{
    const float y = c.y;          // (bit_cast applied to the element expression itself reads element 0)
    return __builtin_bit_cast(int, y);
}

struct XrSmem {
    cell_t *em, *al, *be;
    float *dummy, *rs;
    int *cnt, *lab, *nxt, *dup, *inv;
    __device__ XrSmem(float *base, int T, int SP, int C)
    {
        cell_t *lat = reinterpret_cast<cell_t *>(base);
        em = lat + kPrefetch * SP;                          // zero pad rows on both sides
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        dummy = reinterpret_cast<float *>(be + (size_t)T * SP);   // [0..1] spare cell, [7] tables-ready flag
        cnt = reinterpret_cast<int *>(dummy + 8);
        lab = cnt + 16;
        nxt = lab + SP;
        dup = nxt + SP;
        inv = dup + SP;
        rs = reinterpret_cast<float *>(inv + C + 4);         // [workers][slots]: 1/sum of the rows (DUAL build)
    }
    // the scaled posterior of row t overwrites the first half of that row's beta cells (same
    // worker wave, after it has read them)
    __device__ __forceinline__ float *grow(int t, int SP) const { return reinterpret_cast<float *>(be + (size_t)t * SP); }
};

static size_t xr_smem_bytes(int T, int SP, int C)
{
    return (size_t)(3 * T + 2 * kPrefetch) * SP * 8 + noblank_tables_bytes(SP, C) + kPipeWorkers * kPipeRows * 4;
}

// neighbour state through DPP, lanes without a neighbour read 0 (= no mass, exponent -inf)
template <bool FWD>
__device__ __forceinline__ int xr_nb(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, FWD ? 0x138 : 0x130, 0xf, 0xf, true);
}

// alpha (FWD) / beta (!FWD) chain over rows that are still being produced; the hand-off
// protocol is the one of lattice_chain_sync (noblank_pipe.hpp).  alpha_t is stored WITH the
// emission of step t, beta_t WITHOUT it (beta_{T_b-1} = [l = L-1]), so that
// gamma_t(l) ~ alpha_t(l) * beta_t(l) needs no emission in P3.  The beta step for row t therefore
// multiplies by the emissions of row t+1 first and adds the neighbours afterwards.
template <bool FWD>
__device__ __forceinline__ cell_t xr_chain_sync(const NoblankParams &p, const cell_t *em, cell_t *out, cell_t *dummy,
                                                const int *cnt, int T, int Tb, int L, int SP)
{
    const int lane = lane_id();
    const bool act = lane < SP;
    const int dir = FWD ? SP : -SP;
    const int t_first = FWD ? 0 : Tb - 1;
    const cell_t *rd = act ? em + t_first * SP + lane : em - kPrefetch * SP;   // idle lanes: a zero pad cell
    cell_t *wr = act ? out + t_first * SP + lane : dummy;
    const int winc = act ? dir : 0;
    float m;
    int k;
    cell_t ring[kPrefetch];

    const int H = (T + 1) >> 1;
    const int pos0 = FWD ? 0 : T - Tb;
    const int own = FWD ? H : T - H;
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    const int *cp = cnt + (lane < kPipeWorkers ? lane : 0);
    int seen = *(lds_cvint *)cp;
    bool starved = false;                                    // a bounded wait ran out: NaN instead of a plausible number
    auto wait_upto = [&](int i_last) {
        const int q = pos0 + (i_last < Tb ? i_last : Tb - 1);
        int need = 0;
        if (q >= own) need = kPipeRows;
        else if (q >= lane) need = 2 * ((q - lane) / kPipeWorkers) + (FWD ? 1 : 2);
        if (lane >= kPipeWorkers) need = 0;
        int spins = 0;
        while (__builtin_amdgcn_ballot_w64(seen < need) != 0) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(1);
            seen = *(lds_cvint *)cp;
        }
        lds_order();
        seen = *(lds_cvint *)cp;
    };
    auto merge = [&]() {                                     // (m,k) += neighbour, in the larger exponent
        const int nk = xr_nb<FWD>(k);
        const float nm = __builtin_bit_cast(float, xr_nb<FWD>(__builtin_bit_cast(int, m)));
        const int kk = k > nk ? k : nk;
        m = __builtin_amdgcn_ldexpf(m, k - kk) + __builtin_amdgcn_ldexpf(nm, nk - kk);
        k = kk;
    };
    auto renorm = [&]() {                                    // mantissa back into [0.5, 1)
        k += __builtin_amdgcn_frexp_expf(m);
        m = __builtin_amdgcn_frexp_mantf(m);
    };
    auto step = [&](cell_t e, bool norm) {
        if (FWD) {
            merge();
            m *= e.x;
            k += cell_k(e);
            if (norm) renorm();
        } else {
            m *= e.x;
            k += cell_k(e);
            if (norm) renorm();
            merge();
        }
        *wr = make_cell(m, k);
        wr += winc;
    };

    int *prog = const_cast<int *>(cnt) + kPipeWorkers + (FWD ? 0 : 1);
    __builtin_amdgcn_s_setprio(3);
    wait_upto(kPrefetch);
    if (FWD) {                                               // alpha_0 = p_0(0) on state 0 only
        const cell_t e0 = *rd;
        rd += winc;
        m = lane == 0 ? e0.x : 0.f;
        k = lane == 0 ? kXrBias + cell_k(e0) : 0;
    } else {                                                 // beta_{T_b-1} = 1 on state L-1 only
        m = lane == L - 1 ? 1.f : 0.f;
        k = lane == L - 1 ? kXrBias : 0;
    }
    *wr = make_cell(m, k);
    wr += winc;
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) { ring[j] = *rd; rd += winc; }
    int i = 1;
    for (; i + kBlockSteps <= Tb; i += kBlockSteps) {
        lds_order();
        *prog = i;
        wait_upto(i + kBlockSteps - 1 + kPrefetch);
        if (CTC_DIAG(p) < 0) stamp(p, 2 + i / kBlockSteps);       // diagnostic: block starts -> slots 2..10
#pragma unroll
        for (int j = 0; j < kBlockSteps; ++j) {
            const cell_t e = ring[j % kPrefetch];
            ring[j % kPrefetch] = *rd;
            rd += winc;
            step(e, j % 4 == 3);
        }
    }
    lds_order();
    *prog = i;
    wait_upto(Tb - 1);
#pragma unroll
    for (int j = 0; j < kBlockSteps; ++j)
        if (i + j < Tb) {
            const cell_t e = ring[j % kPrefetch];
            ring[j % kPrefetch] = *rd;                       // may run past the last row: zero pad / unused
            rd += winc;
            step(e, j % 4 == 3);
        }
    lds_order();
    *prog = Tb;
    __builtin_amdgcn_s_setprio(0);
    if (starved) {
        raise_status(p.counter, kStatusNoblankStarved);
        return make_cell(__builtin_nanf(""), 0);
    }
    return make_cell(m, k);
}

template <int CH, bool DUAL>
__global__ __launch_bounds__(kThreads, DUAL ? 8 : 4) void noblank_xr_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    const XrSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.C);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64 * kChainB) note_arrival(p.counter, b);        // (the beta chain wave has no other vector-memory operation)
    const int u = (w == 0 || w == kChainB) ? -1 : (w < kChainB ? w - 1 : w - 2);
    const float ninf = -__builtin_inff();

    stamp(p, 0);
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const int raw_label = tid < p.S ? load_label(p.lab, p.lab64, (int64_t)b * p.S + tid) : 0;
    float v[kPipeRows][CH];
    if (u >= 0) {
#pragma unroll
        for (int r = 0; r < kPipeRows; ++r) {
            const int t = pipe_row(p.T, u, r);
            const float *row = row_ptr(p, t >= 0 ? t : 0, b);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = lane + 64 * j;
                v[r][j] = row[c < p.C ? c : p.C - 1];
            }
        }
    }
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
    if (tid < 16) sm.cnt[tid] = 0;
    const cell_t zero = make_cell(0.f, 0);
    for (int i = tid; i < kPrefetch * p.SP; i += kThreads) {
        sm.em[i - kPrefetch * p.SP] = zero;
        sm.em[p.T * p.SP + i] = zero;
    }
    if (tid < 8) sm.dummy[tid] = 0.f;
    __syncthreads();
    stamp(p, 1);

    // ---------------------------------------------------------------- chain waves
    if (u < 0) {
        if (Tb > 0) {
            cell_t *spare = reinterpret_cast<cell_t *>(sm.dummy);
            if (w == 0) {
                const cell_t a = xr_chain_sync<true>(p, sm.em, sm.al, spare, sm.cnt, p.T, Tb, L, p.SP);
                stamp(p, 11);
                // nll = -log alpha[T_b-1, L_b-1] (NoBlankCTC.py:58-68,139) from lane L-1
                const float am = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a.x), L - 1));
                const int ak = __builtin_amdgcn_readlane(cell_k(a), L - 1);
                float nll = am > 0.f ? -(__builtin_amdgcn_logf(am) + (float)(ak - kXrBias)) * kLn2 : -kNeg;
                if (am != am) nll = am;                      // starved hand-off
                publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
            } else if (p.grad) {
                xr_chain_sync<false>(p, sm.em, sm.be, spare, sm.cnt, p.T, Tb, L, p.SP);
                stamp(p, 11);
            }
        } else if (w == 0) {
            publish_and_reduce_sum(-kNeg, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        }
        return;
    }

    // ---------------------------------------------------------------- workers
    constexpr int kGroup = 4;
    const float mask_tail = lane + 64 * (CH - 1) < p.C ? 0.f : ninf;
    const int lab_l = lane < p.SP ? sm.lab[lane] : 0;
    const int lab_src = lab_l & 63, lab_chunk = lab_l >> 6;
    // grad_scale / sum_c exp(x - max) per slot: registers with one workgroup per CU; in the 64-VGPR
    // build they go to LDS (12 registers fewer: without that the build spills 10 VGPRs to scratch,
    // i.e. to HBM -- measured +40 % memory traffic at B = 2048)
    float rsrow[DUAL ? 1 : kPipeRows];
    float *const rs_lds = sm.rs + u * kPipeRows;
#pragma unroll
    for (int gq = 0; gq < kPipeRows / kGroup; ++gq) {        // P1: extremes first
        float m[kGroup], sum[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int r = gq * kGroup + k;
            m[k] = v[r][CH - 1] + mask_tail;
#pragma unroll
            for (int j = 0; j < CH - 1; ++j) m[k] = fmaxf(m[k], v[r][j]);
        }
        wave_max4(m[0], m[1], m[2], m[3]);
        float xv[kGroup];
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int r = gq * kGroup + k;
            xv[k] = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float q = __shfl(v[r][j], lab_src, kWave);
                if (lab_chunk == j) xv[k] = q;
            }
            const float mb = -m[k] * kLog2e;
            v[r][CH - 1] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[r][CH - 1] + mask_tail, kLog2e, mb));
            sum[k] = v[r][CH - 1];
#pragma unroll
            for (int j = 0; j < CH - 1; ++j) {
                v[r][j] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[r][j], kLog2e, mb));
                sum[k] += v[r][j];
            }
        }
        wave_sum4(sum[0], sum[1], sum[2], sum[3]);
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            const int r = gq * kGroup + k;
            const int t = pipe_row(p.T, u, r);
            const float l2sum = __builtin_amdgcn_logf(sum[k]);
            if (DUAL) {
                rs_lds[r] = p.grad_scale * __builtin_amdgcn_rcpf(sum[k]);    // (every lane, same value)
            } else {
                rsrow[r] = p.grad_scale * __builtin_amdgcn_rcpf(sum[k]);
                asm volatile("" : "+v"(rsrow[r]));
            }
            // emission e = log_softmax(x)[lab_l] in log2 units, split into 2^floor * 2^frac
            const float e2 = fmaxf(__builtin_fmaf(xv[k] - m[k], kLog2e, -l2sum), kXrMinLog2);
            const float fl = __builtin_floorf(e2);
            const float pm = __builtin_amdgcn_exp2f(e2 - fl);
            if (t >= 0 && t < Tb && lane < p.SP)             // (t: wave-uniform)
                sm.em[t * p.SP + lane] = (lane < L) ? make_cell(pm, (int)fl) : zero;
            lds_order();
            if (lane == 0) sm.cnt[u] = r + 1;
        }
    }
    stamp(p, 2);
    if (!p.grad) return;

    if (u == kPipeWorkers - 1) {
        for (int c = lane; c < p.C; c += kWave) sm.inv[c] = 0x7fffffff;
        int k = 0, n = -1;
        if (lane < L) {
            k = sm.lab[lane];
            atomicMin(&sm.inv[k], lane);
            for (int l2 = lane + 1; l2 < L; ++l2)
                if (sm.lab[l2] == k) { n = l2; break; }
        }
        if (lane < p.SP) sm.nxt[lane] = n;
        for (int c = lane; c < p.C; c += kWave)
            if (sm.inv[c] == 0x7fffffff) sm.inv[c] = -1;
        if (lane < p.SP) sm.dup[lane] = (lane < L && sm.inv[k] == lane && n >= 0) ? 1 : 0;
        lds_order();
        if (lane == 0) sm.dummy[7] = 1.0f;
    }
    bool starved = false;                                    // a bounded wait ran out: NaN gradient rows, status raised
    {
        int spins = 0;
        while (lds_now(sm.dummy + 7) == 0.f) {
            if (++spins >= kSpinLimit) { starved = true; break; }
            __builtin_amdgcn_s_sleep(4);
        }
        lds_order();
    }
    int first[CH];
    bool has[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int c = lane + 64 * j;
        const int f = (c < p.C) ? sm.inv[c] : -1;
        has[j] = f >= 0;
        first[j] = f >= 0 ? f : 0;
    }
    const int my_dup = lane < p.SP ? sm.dup[lane] : 0;
    const bool any_dup = __builtin_amdgcn_ballot_w64(my_dup != 0) != 0;
    const int Tlive = Tb;
    const int lcl = lane < p.SP ? lane : 0;
    const float gsc0 = p.grad_scale;
    int shiftk = 0;                                          // shared exponent shift of the posterior rows
    bool have_shift = false;
    const bool pair = p.SP <= 32;
    const bool upper = lane >= 32;
    const int hl = lane & 31, hcl = hl < p.SP ? hl : 0;
    const int my_dup_h = hl < p.SP ? sm.dup[hl] : 0;
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;

#pragma unroll
    for (int gq = kPipeRows / kGroup - 1; gq >= 0; --gq) {
        int tt[kGroup];
        int need_a = 0, need_b = 0;
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            tt[k] = pipe_row(p.T, u, gq * kGroup + k);
            if (tt[k] >= 0 && tt[k] < Tlive) {
                need_a = max(need_a, tt[k] + 1);
                need_b = max(need_b, Tlive - tt[k]);
            }
        }
        if (need_a > 0) {
            int spins = 0;
            while (*(lds_cvint *)(sm.cnt + kPipeWorkers) < need_a || *(lds_cvint *)(sm.cnt + kPipeWorkers + 1) < need_b) {
                if (++spins >= kSpinLimit) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();
        }
        if (starved) raise_status(p.counter, kStatusNoblankStarved);
        const float gsc = starved ? __builtin_nanf("") : gsc0;
        if (CTC_DIAG(p) < 0) stamp(p, 3 + (2 - gq));
        // gamma_t(l) = alpha_t(l) beta_t(l) / sum_l' (...): products of mantissas, exponents added;
        // every row's total is the same number P(x, labels) up to rounding, so the exponent of
        // the largest cell of the first (middle) group shifts all later rows into range too.
        if (pair) {
            float pr[2], z[2];
            int ks[2], tl[2];
            bool lvl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                tl[q] = upper ? tt[2 * q + 1] : tt[2 * q];
                lvl[q] = tl[q] >= 0 && tl[q] < Tlive;
                const int off = (lvl[q] ? tl[q] : 0) * p.SP + hcl;
                const cell_t a = sm.al[off], bb = sm.be[off];
                pr[q] = (hl < p.SP && lvl[q]) ? a.x * bb.x : 0.f;
                ks[q] = cell_k(a) + cell_k(bb);
            }
            if (!have_shift && need_a > 0) {
                float e0 = pr[0] > 0.f ? (float)(ks[0] + __builtin_amdgcn_frexp_expf(pr[0])) : ninf;
                float e1 = pr[1] > 0.f ? (float)(ks[1] + __builtin_amdgcn_frexp_expf(pr[1])) : ninf;
                halves_max2(e0, e1, upper);
                float mx = fmaxf(e0, e1);
                mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));
                shiftk = mx > ninf ? (int)mx : 0;
                have_shift = true;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pr[q] = __builtin_amdgcn_ldexpf(pr[q], ks[q] - shiftk);
                z[q] = pr[q];
            }
            halves_sum2(z[0], z[1], upper);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float *gr = sm.grow(lvl[q] ? tl[q] : 0, p.SP);
                if (lvl[q] && hl < p.SP) gr[hl] = pr[q] * (gsc * __builtin_amdgcn_rcpf(z[q]));
                if (any_dup && lvl[q] && my_dup_h) {
                    float tot = gr[hl];
                    for (int n = sm.nxt[hl]; n >= 0; n = sm.nxt[n]) tot += gr[n];
                    gr[hl] = tot;
                }
            }
        } else {
            float pr[kGroup], z[kGroup];
            int ks[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                const bool lv = tt[k] >= 0 && tt[k] < Tlive;
                const int off = (lv ? tt[k] : 0) * p.SP + lcl;
                const cell_t a = sm.al[off], bb = sm.be[off];
                pr[k] = (lane < p.SP && lv) ? a.x * bb.x : 0.f;
                ks[k] = cell_k(a) + cell_k(bb);
            }
            if (!have_shift && need_a > 0) {
                float e[kGroup];
#pragma unroll
                for (int k = 0; k < kGroup; ++k)
                    e[k] = pr[k] > 0.f ? (float)(ks[k] + __builtin_amdgcn_frexp_expf(pr[k])) : ninf;
                wave_max4(e[0], e[1], e[2], e[3]);
                const float mx = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
                shiftk = mx > ninf ? (int)mx : 0;
                have_shift = true;
            }
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                pr[k] = __builtin_amdgcn_ldexpf(pr[k], ks[k] - shiftk);
                z[k] = pr[k];
            }
            wave_sum4(z[0], z[1], z[2], z[3]);
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                const int t = tt[k];
                if (t < 0 || t >= Tlive) continue;
                float *gr = sm.grow(t, p.SP);
                if (lane < p.SP) gr[lane] = pr[k] * (gsc * __builtin_amdgcn_rcpf(z[k]));
                if (any_dup && my_dup) {
                    float tot = gr[lane];
                    for (int n = sm.nxt[lane]; n >= 0; n = sm.nxt[n]) tot += gr[n];
                    gr[lane] = tot;
                }
            }
        }
#pragma unroll
        for (int k = kGroup - 1; k >= 0; --k) {
            const int r = gq * kGroup + k, t = tt[k];
            if (t < 0) continue;
            float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
            if (t >= Tlive) {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    if (j < CH - 1 || c < p.C) stream_store(&g[c], 0.f);
                }
                continue;
            }
            const float *gr = sm.grow(t, p.SP);
            const float rsr = DUAL ? rs_lds[r] : rsrow[r];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = lane + 64 * j;
                const float occ = gr[first[j]];
                const float gv = __builtin_fmaf(v[r][j], rsr, has[j] ? -occ : 0.f);
                if (j < CH - 1 || c < p.C) stream_store(&g[c], gv);
            }
        }
    }
    stamp(p, 7);
}

}  // namespace ctc
