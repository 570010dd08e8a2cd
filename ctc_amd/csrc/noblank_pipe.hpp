// Pipelined variant of the fused no-blank kernel (included by noblank.hip).
//
// Same arithmetic as noblank_fused_kernel, different schedule: at B = #CUs every CU runs one
// workgroup and all of them would walk load -> chains -> gradient in lock-step, leaving HBM
// idle while the chains run and the ALUs idle while HBM streams.  Here the three overlap
// INSIDE the workgroup:
//
//   waves 0,1   dedicated chain waves (alpha, beta').  They start right after the setup
//               barrier and consume emission rows as soon as a worker has published them.
//   waves 2..15 14 workers, kPipeRows row slots each.  Slots alternate front / back of the
//               sequence (slot 0 = rows 0..13, slot 1 = rows T-1..T-14, slot 2 = rows 14..27,
//               ...) and are loaded / emitted in that order, so alpha's first rows and beta's
//               first rows arrive first and the chains run just behind the HBM stream.  A
//               worker then turns its rows into gradient in the opposite order (middle of the
//               sequence first): row t is ready once alpha AND beta' have passed it, i.e.
//               middle-out, so the stores overlap the second half of the chains.
//
// Hand-off through LDS counters only (a wave's LDS operations complete in order, so a counter
// store behind a row store publishes the row).  Workers -> chains: a per-worker count of
// published row slots, looked at once per kBlockSteps chain steps (closed-form requirement
// per worker), so the steps themselves stay branch-free.  Chains -> workers: each chain
// publishes how many steps it has completed, once per block; a worker looks once per group of
// four slots and sleeps in between, so it does not steal issue slots from the chain waves
// (which also run at raised priority).  Producers never wait on consumers, so there is no
// cycle; every spin is bounded (kSpinLimit) so the grid drains even if a hand-off were broken.
#pragma once

namespace ctc {

constexpr int kPipeWorkers = kWaves - 2;
#ifndef CTC_CHAIN_B
#define CTC_CHAIN_B 2
#endif
constexpr int kChainB = CTC_CHAIN_B;   // wave of the beta' chain: 2 measured 0.4 us faster than 1, 4 slower
constexpr int kPipeRows = 12;                               // slots per worker
constexpr int kPipeMaxT = kPipeWorkers * kPipeRows;         // 168
constexpr int kSpinLimit = 1 << 20;

// a re-read that must really go to LDS again (another wave writes the row): volatile, but in
// the LDS address space -- a generic volatile load would become a flat_load sc0 sc1 + vmcnt(0)
typedef const volatile __attribute__((address_space(3))) float lds_cvfloat;
__device__ __forceinline__ float lds_now(const float *p) { return *(lds_cvfloat *)p; }

// row owned by worker u in slot r (or -1): even slots walk the front half upwards, odd slots
// the back half downwards
__device__ __forceinline__ int pipe_row(int T, int u, int r)
{
    const int H = (T + 1) >> 1, idx = u + kPipeWorkers * (r >> 1);
    if ((r & 1) == 0) return idx < H ? idx : -1;
    return idx < T - H ? T - 1 - idx : -1;
}

constexpr int kBlockSteps = 16;  // chain steps between two looks at the workers' progress

// alpha / beta' chain over rows that are still being produced (K = 1: S <= 64).
// `cnt[u]` = number of row slots worker u has published (slots are published in order).  A
// chain walks its own half of the sequence in exactly the order the workers publish it, so
// "rows up to position q are there" is a closed form per worker; once the chain crosses into
// the other half it simply requires every slot.  The look happens once per kBlockSteps steps,
// kPrefetch rows ahead, so the steps in between are branch-free (exact lgkmcnt waits).
template <bool FWD, bool ROT>
__device__ __forceinline__ float lattice_chain_sync(const NoblankParams &p, const float *em, float *out, float *dummy,
                                                    const int *cnt, int T, int Tb, int L, int SP)
{
    const int lane = lane_id();
    const bool act = lane < SP;
    const int dir = FWD ? SP : -SP;
    const int t_first = FWD ? 0 : Tb - 1;
    const float *rd = act ? em + t_first * SP + lane : em - kPrefetch * SP;   // idle lanes: sentinel pad
    float *wr = act ? out + t_first * SP + lane : dummy;
    const int winc = act ? dir : 0;
    float a;
    float ring[kPrefetch];

    const int H = (T + 1) >> 1;
    const int pos0 = FWD ? 0 : T - Tb;                       // position (within the half order) of step 0
    const int own = FWD ? H : T - H;                         // positions below this are the near half
    // The counters are read one block AHEAD (they only grow, so a stale value that already
    // satisfies the requirement is as good as a fresh one): the read issued at the start of
    // a block is consumed at the start of the next, its latency hidden behind the steps.
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;
    const int *cp = cnt + (lane < kPipeWorkers ? lane : 0);
    int seen = *(lds_cvint *)cp;
    bool starved = false;                                    // a bounded wait ran out: NaN instead of a plausible number
    auto wait_upto = [&](int i_last) {                       // rows of steps 0..i_last must be published
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
        lds_order();                                         // rows are read only after the look
        seen = *(lds_cvint *)cp;                             // for the next look
    };
    auto shift = [&]() {
        if (ROT)
            return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a),
                                                                      FWD ? 0x13C : 0x134, 0xf, 0xf, false));
        return FWD ? wave_shr1(a, kNeg) : wave_shl1(a, kNeg);
    };
    auto step = [&](float e) {
        const float adv = shift();
        const float t = __builtin_amdgcn_exp2f(-fabsf(a - adv) * kLog2e);
        a = __builtin_fmaf(__builtin_amdgcn_logf(1.0f + t), kLn2, vmax(a, adv)) + e;
        *wr = a;
        wr += winc;
    };

    // progress for the workers: steps completed (their row stores are older LDS ops of this wave)
    int *prog = const_cast<int *>(cnt) + kPipeWorkers + (FWD ? 0 : 1);
    __builtin_amdgcn_s_setprio(3);                           // the chains are the critical path
    wait_upto(kPrefetch);
    a = (lane == (FWD ? 0 : L - 1)) ? *rd : kNeg;            // first row: only "stay" from the start state
    *wr = a;
    wr += winc;
    rd += winc;
#pragma unroll
    for (int j = 0; j < kPrefetch; ++j) { ring[j] = *rd; rd += winc; }
    int i = 1;
    for (; i + kBlockSteps <= Tb; i += kBlockSteps) {
        lds_order();                                         // row stores above, then the count
        *prog = i;                                           // (every lane, same value: no exec juggling)
        wait_upto(i + kBlockSteps - 1 + kPrefetch);
#pragma unroll
        for (int j = 0; j < kBlockSteps; ++j) {
            const float e = ring[j % kPrefetch];
            ring[j % kPrefetch] = *rd;
            rd += winc;
            step(e);
        }
    }
    lds_order();
    *prog = i;
    wait_upto(Tb - 1);
#pragma unroll
    for (int j = 0; j < kBlockSteps; ++j)
        if (i + j < Tb) {
            const float e = ring[j % kPrefetch];
            ring[j % kPrefetch] = *rd;                       // may run past the last row: pad / unused
            rd += winc;
            step(e);
        }
    lds_order();
    *prog = Tb;
    __builtin_amdgcn_s_setprio(0);
    if (starved) {
        raise_status(p.counter, kStatusNoblankStarved);
        return __builtin_nanf("");
    }
    return a;
}

// DUAL: cap the kernel at 64 VGPRs (8 waves/SIMD) so that TWO 16-wave workgroups share a CU.
// With more samples than CUs their phases then interleave (one streams while the other
// scans); with one workgroup per CU the uncapped build (66 VGPRs, no scratch) is faster.
template <int CH, bool DUAL>
__global__ __launch_bounds__(kThreads, DUAL ? 8 : 4) void noblank_pipelined_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    const NoblankSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.C);
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id(), lane = lane_id();
    if (tid == 64 * kChainB) note_arrival(p.counter, b);        // (the beta chain wave has no other vector-memory operation)
    // roles: waves 0 and kChainB are the chains, the rest are workers 0..13
    const int u = (w == 0 || w == kChainB) ? -1 : (w < kChainB ? w - 1 : w - 2);
    const float ninf = -__builtin_inff();

    stamp(p, 0);
    // loads first (oldest = needed first): lengths (scalar), this thread's label, the rows
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

    // setup: label table, marker rows, pads; ONE barrier, then the roles diverge
    if (tid < p.SP) {
        int k = 0;
        if (tid < L) {
            k = raw_label % p.C;
            if (k < 0) k += p.C;                             // python negative index (NoBlankCTC.py:102)
        }
        sm.lab[tid] = k;
    }
    if (tid < 16) sm.cnt[tid] = 0;
    for (int i = tid; i < kPrefetch * p.SP; i += kThreads) {
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }
    if (tid < 8) sm.dummy[tid] = 0.f;
    if (tid == 8) sm.dummy[7] = 0.f;
    __syncthreads();
    stamp(p, 1);

    // ---------------------------------------------------------------- chain waves
    if (u < 0) {
        if (Tb > 0) {
            const bool rot = p.SP <= 63;
            if (w == 0) {
                const float a = rot ? lattice_chain_sync<true, true>(p, sm.em, sm.al, sm.dummy, sm.cnt, p.T, Tb, L, p.SP)
                                    : lattice_chain_sync<true, false>(p, sm.em, sm.al, sm.dummy, sm.cnt, p.T, Tb, L, p.SP);
                stamp(p, 2);
                // nll = -alpha[T_b-1, L_b-1] (NoBlankCTC.py:58-68,139) straight from lane L-1
                const float nll = -__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), L - 1));
                publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
            } else if (p.grad) {                             // w == kChainB
                if (rot) lattice_chain_sync<false, true>(p, sm.em, sm.be, sm.dummy, sm.cnt, p.T, Tb, L, p.SP);
                else lattice_chain_sync<false, false>(p, sm.em, sm.be, sm.dummy, sm.cnt, p.T, Tb, L, p.SP);
                stamp(p, 2);
            }
        } else if (w == 0) {
            publish_and_reduce_sum(-kNeg, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
        }
        stamp(p, 7);
        return;
    }

    // ---------------------------------------------------------------- workers
    // Four row slots are processed in lock-step (wave_max4 / wave_sum4 interleave their
    // reductions); a slot without a live row computes on harmless duplicate data and is
    // only kept away from the stores.
    constexpr int kGroup = 4;
    const float mask_tail = lane + 64 * (CH - 1) < p.C ? 0.f : ninf;   // only the last chunk is partial
    const int lab_l = lane < p.SP ? sm.lab[lane] : 0;
    const int lab_src = lab_l & 63, lab_chunk = lab_l >> 6;
    float rsrow[kPipeRows];                                  // grad_scale / sum_c exp(x - max) per slot
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
            for (int j = 0; j < CH; ++j) {                   // gather x[t, lab_l] out of the row registers
                const float q = __shfl(v[r][j], lab_src, kWave);
                if (lab_chunk == j) xv[k] = q;
            }
            // the row registers now become exp(x - max): the gradient pass needs softmax(x) =
            // that times 1/sum, so no second exp per element; the emission above used x itself
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
            const float lsum = fast_log(sum[k]);
            rsrow[r] = p.grad_scale * __builtin_amdgcn_rcpf(sum[k]);   // grad_scale / sum_c exp(x - max)
            asm volatile("" : "+v"(rsrow[r]));               // keep it in a VGPR: 12 wave-uniform floats
                                                             // would spill the scalar file
            if (t >= 0 && t < Tb && lane < p.SP)             // (t: wave-uniform)
                sm.em[t * p.SP + lane] = (lane < L) ? (xv[k] - m[k]) - lsum : kNeg;
            lds_order();
            if (lane == 0) sm.cnt[u] = r + 1;                // publishes the slot (same wave: in order)
        }
    }
    stamp(p, 2);
    if (!p.grad) return;

    // the last worker builds the class tables of P3 (after its rows are out: the chains wait
    // for those, nobody waits for the tables before P3)
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
        if (lane == 0) sm.dummy[7] = 1.0f;                   // tables ready (same wave: LDS stores in order)
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
        first[j] = f >= 0 ? f : 0;                           // clamped: the read is unconditional
    }
    const int my_dup = lane < p.SP ? sm.dup[lane] : 0;
    const bool any_dup = __builtin_amdgcn_ballot_w64(my_dup != 0) != 0;   // wave-uniform: most samples have none
    const int Tlive = Tb;                                    // ok <=> an alignment exists (L_b <= T_b)
    const int lcl = lane < p.SP ? lane : 0;
    const bool in = lane < L;
    const float gsc0 = p.grad_scale;
    float shift = 0.f;                                       // row shift of the posterior softmax
    bool have_shift = false;
    const bool pair = p.SP <= 32;                            // two posterior rows per wave pass
    const bool upper = lane >= 32;
    const int hl = lane & 31, hcl = hl < p.SP ? hl : 0;
    const int my_dup_h = hl < p.SP ? sm.dup[hl] : 0;
    typedef const volatile __attribute__((address_space(3))) int lds_cvint;

    // P3: middle-out, one look at the chains' progress per group of four slots
#pragma unroll
    for (int gq = kPipeRows / kGroup - 1; gq >= 0; --gq) {
        int tt[kGroup];
        int need_a = 0, need_b = 0;
#pragma unroll
        for (int k = 0; k < kGroup; ++k) {
            tt[k] = pipe_row(p.T, u, gq * kGroup + k);
            if (tt[k] >= 0 && tt[k] < Tlive) {               // alpha must have passed the largest t,
                need_a = max(need_a, tt[k] + 1);             // beta' (walking down from T_b-1) the smallest
                need_b = max(need_b, Tlive - tt[k]);
            }
        }
        if (need_a > 0) {
            int spins = 0;
            while (*(lds_cvint *)(sm.cnt + kPipeWorkers) < need_a || *(lds_cvint *)(sm.cnt + kPipeWorkers + 1) < need_b) {
                if (++spins >= kSpinLimit) { starved = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            lds_order();                                     // lattice rows are read only after the look
        }
        if (starved) raise_status(p.counter, kStatusNoblankStarved);
        const float gsc = starved ? __builtin_nanf("") : gsc0;
        if (CTC_DIAG(p) < 0) stamp(p, 3 + (2 - gq));             // diagnostic: groups 2,1,0 -> slots 3,4,5
        // gamma_t = softmax_l(alpha_t + beta'_t - e_t): row-normalised posterior (lattice.hpp).
        // Any shift gives the same softmax; every row's log-normaliser equals -nll up to the
        // rounding of the two scans, so the maximum found for the first (middle) group serves all
        // later rows too: z - shift stays <= ~0, and the row sum still normalises exactly.
        if (pair) {
            // S <= 32: two rows side by side in the two 32-lane halves -> half the LDS reads,
            // exps and reduction steps.  Slots (0,1) and (2,3) of the group form the two pairs.
            float zp[2], pp[2];
            int tl[2];
            bool lvl[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                tl[q] = upper ? tt[2 * q + 1] : tt[2 * q];
                lvl[q] = tl[q] >= 0 && tl[q] < Tlive;        // per half
                const int off = (lvl[q] ? tl[q] : 0) * p.SP + hcl;
                const float zz = sm.al[off] + sm.be[off] - sm.em[off];
                zp[q] = (hl < L && lvl[q]) ? zz : ninf;
                pp[q] = zp[q];
            }
            if (!have_shift && need_a > 0) {
                halves_max2(zp[0], zp[1], upper);
                float mx = fmaxf(zp[0], zp[1]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, kWave));   // over both halves
                shift = mx;
                have_shift = true;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                pp[q] = __builtin_amdgcn_exp2f((pp[q] - shift) * kLog2e);    // exp2(-inf) = 0 beyond L
                zp[q] = pp[q];
            }
            halves_sum2(zp[0], zp[1], upper);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int off = (lvl[q] ? tl[q] : 0) * p.SP;
                // gamma * grad_scale goes to LDS, so the dense row is one fma and one subtract
                if (lvl[q] && hl < p.SP) sm.be[off + hl] = pp[q] * (gsc * __builtin_amdgcn_rcpf(zp[q]));
                if (any_dup && lvl[q] && my_dup_h) {         // fold repeats onto the first occurrence
                    float tot = sm.be[off + hl];
                    for (int n = sm.nxt[hl]; n >= 0; n = sm.nxt[n]) tot += sm.be[off + n];
                    sm.be[off + hl] = tot;
                }
            }
        } else {
            float z[kGroup], pe[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                const bool lv = tt[k] >= 0 && tt[k] < Tlive;       // wave-uniform; idle slots read row 0 and
                const int off = (lv ? tt[k] : 0) * p.SP + lcl;     // are masked (row 0 may not exist yet)
                const float zz = sm.al[off] + sm.be[off] - sm.em[off];
                z[k] = (in && lv) ? zz : ninf;
                pe[k] = z[k];
            }
            if (!have_shift && need_a > 0) {                 // first group with a live row (wave-uniform)
                wave_max4(z[0], z[1], z[2], z[3]);
                shift = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));     // idle slots contribute -inf
                have_shift = true;
            }
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                pe[k] = __builtin_amdgcn_exp2f((pe[k] - shift) * kLog2e);    // exp2(-inf) = 0 beyond L
                z[k] = pe[k];
            }
            wave_sum4(z[0], z[1], z[2], z[3]);
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                const int t = tt[k];
                if (t < 0 || t >= Tlive) continue;           // wave-uniform
                const int off = t * p.SP;
                if (lane < p.SP) sm.be[off + lane] = pe[k] * (gsc * __builtin_amdgcn_rcpf(z[k]));
                if (any_dup && my_dup) {
                    float tot = sm.be[off + lane];
                    for (int n = sm.nxt[lane]; n >= 0; n = sm.nxt[n]) tot += sm.be[off + n];
                    sm.be[off + lane] = tot;
                }
            }
        }
        // dense rows: grad = softmax(x) * scale - gamma' gathered by class
#pragma unroll
        for (int k = kGroup - 1; k >= 0; --k) {
            const int r = gq * kGroup + k, t = tt[k];
            if (t < 0) continue;                             // wave-uniform
            float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
            if (t >= Tlive) {                                // wave-uniform: dead row
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    if (j < CH - 1 || c < p.C) stream_store(&g[c], 0.f);
                }
                continue;
            }
            const int off = t * p.SP;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = lane + 64 * j;
                const float occ = sm.be[off + first[j]];
                const float gv = __builtin_fmaf(v[r][j], rsrow[r], has[j] ? -occ : 0.f);
                if (j < CH - 1 || c < p.C) stream_store(&g[c], gv);
            }
        }
    }
    stamp(p, 7);
}

}  // namespace ctc
