// Fused no-blank CTC loss + input gradient for gfx950 (MI355X).
//
// Replaces NoBlankCTC.forward (NoBlankCTC.py:129-141) and the autograd backward the
// reference runs over it (train.py:444).  One 16-wave workgroup per sample b, one launch:
//
//   P1  at kernel entry every wave issues the loads of its kRows rows x[t,b,:]
//       (coalesced 256-B requests; the rows then STAY in registers until P3).  While
//       they are in flight the workgroup builds the label tables.  Row max / sum-exp
//       (LogSoftmax(dim=2), :136) by DPP reductions; the S emissions
//       e[t,l] = lp[t, lab[l]] (:96-102) are gathered out of the row registers with
//       ds_bpermute and stored to LDS.
//   P2  wave 0 runs the alpha scan, wave 1 the mirrored beta' scan, concurrently,
//       one lattice row per step, states across lanes (lattice.hpp).
//   P3  each wave turns its own rows into gradient: gamma_t = softmax_l(alpha_t +
//       beta'_t - e_t) (row-normalised; repeated labels folded onto the first
//       occurrence), grad = scale * (softmax(x) - gamma scattered by class) from the
//       resident row registers, written with 256-B coalesced stores.
//
// The batch mean is taken in-launch by the last workgroup to publish its nll
// (common.hpp), overlapped with P3.  HBM traffic = read x once, write grad once.
// Shapes beyond the register-resident tiling (T > 160 rows per pass, C > 256) fall
// back to re-reading rows from L2 in P3 / to strided row passes.
#include "lattice.hpp"
#include "launch.hpp"

namespace ctc {

struct NoblankParams {
    const float *x;
    int64_t st, sb;
    const void *lab;
    int lab64;
    const int64_t *in_len, *tgt_len;
    int T, B, C, S, SP;
    int stop;                   // debug: leave after phase `stop` (0 = run everything);
                                // < 0: workgroup 0 / wave -stop-1 stamps (s_memtime, s_memrealtime)
                                // pairs into workspace bytes [64,256) at each phase boundary
    float loss_scale, grad_scale;
    float *nll, *loss, *grad;
    float *gamma;               // optional [B][T][S] posteriors output (ctc_amd_noblank_posteriors)
    float *lattice;             // global-memory lattice slabs (workspace, behind header and list) when T x S exceeds LDS
    int64_t slab;               // floats per sample in `lattice`
    unsigned *counter;
    int next_round;             // > 0: B exceeds one round of workgroups -- blocks prefetch for block + next_round
    float ls_a, ls_b;           // label-smoothed emission a lp[c_l] + b sum_n lp[n] (NoBlankCTC.py:100-107); 1, 0: plain
    int koff;                   // noblank_km_kernel: what the exponent recurrence starts with (log2 of the path count / 2)
};

// Every field through an empty volatile asm: the copy holds the same values, but the compiler no longer knows that
// they are the launch's constants.  For kernels that LOOP over samples: whatever is computed from `p` inside the loop
// stays inside (hoisted out, the address arithmetic of a whole sample waits in registers for its turn -- see
// noblank_r16_kernel<.., PS>).
__device__ __forceinline__ NoblankParams opaque_params(NoblankParams q)
{
#define CTC_OPQ(f) asm volatile("" : "+s"(q.f))
    CTC_OPQ(x); CTC_OPQ(st); CTC_OPQ(sb); CTC_OPQ(lab); CTC_OPQ(lab64); CTC_OPQ(in_len); CTC_OPQ(tgt_len);
    CTC_OPQ(T); CTC_OPQ(B); CTC_OPQ(C); CTC_OPQ(S); CTC_OPQ(SP); CTC_OPQ(stop); CTC_OPQ(loss_scale); CTC_OPQ(grad_scale);
    CTC_OPQ(nll); CTC_OPQ(loss); CTC_OPQ(grad); CTC_OPQ(gamma); CTC_OPQ(lattice); CTC_OPQ(slab); CTC_OPQ(counter);
    CTC_OPQ(next_round); CTC_OPQ(ls_a); CTC_OPQ(ls_b); CTC_OPQ(koff);
#undef CTC_OPQ
    return q;
}

#ifndef CTC_NOBLANK_THREADS
#define CTC_NOBLANK_THREADS 1024
#endif
constexpr int kThreads = CTC_NOBLANK_THREADS;  // 16 waves: one pass of kRows rows each covers T <= 160
constexpr int kWaves = kThreads / kWave;
constexpr int kRows = 160 / kWaves;            // rows of x resident per wave
constexpr int kRowsPerPass = kWaves * kRows;

struct NoblankSmem {
    float *em, *al, *be, *mx, *ls, *dummy;
    int *cnt, *lab, *nxt, *dup, *inv;
    // `glb` != nullptr: the T x S arrays live in that global slab (long sequences), only the
    // small tables stay in LDS
    __device__ NoblankSmem(float *base, int T, int SP, int C, float *glb = nullptr)
    {
        float *lat = glb ? glb : base;
        em = lat + kPrefetch * SP;                      // pad rows on both sides (lattice.hpp)
        al = em + (size_t)(T + kPrefetch) * SP;
        be = al + (size_t)T * SP;
        mx = be + (size_t)T * SP;
        ls = mx + T;
        dummy = glb ? base : ls + T;
        cnt = reinterpret_cast<int *>(dummy + 8);            // 16 progress counters (pipelined kernel)
        lab = cnt + 16;
        nxt = lab + SP;
        dup = nxt + SP;
        inv = dup + SP;
    }
};

static size_t noblank_lattice_floats(int T, int SP) { return (size_t)(3 * T + 2 * kPrefetch) * SP + 2 * (size_t)T; }
static size_t noblank_tables_bytes(int SP, int C) { return (8 + 16 + 3 * (size_t)SP + C + 4) * 4; }
static size_t noblank_smem_bytes(int T, int SP, int C)
{
    return noblank_lattice_floats(T, SP) * 4 + noblank_tables_bytes(SP, C);
}
// extra workspace (beyond the first 256 B) of the no-blank entry points: 0 while the lattice
// fits in LDS, else one slab per sample
size_t noblank_extra_workspace(int T, int B, int C, int S)
{
    int K = 1;
    while (K <= 4 && S > kWave * K) K *= 2;
    const int SP = (S + K - 1) / K * K;
    if (K > 4 || noblank_smem_bytes(T, SP, C) <= kMaxLds) return 0;
    return (size_t)B * noblank_lattice_floats(T, SP) * 4;
}

__device__ __forceinline__ const float *row_ptr(const NoblankParams &p, int t, int b)
{
    return p.x + (int64_t)t * p.st + (int64_t)b * p.sb;
}

// ---- register-resident rows (C <= 64*CH) ------------------------------------------
template <int CH>
struct Rows {
    float v[kRows][CH];

    // rows t0..t0+kRows-1.  Every load is issued unconditionally (row / column indices
    // clamped into range) so that the compiler can wait for OLDER loads with an exact
    // vmcnt(N) while these stay in flight; out-of-range values are masked at the use.
    __device__ __forceinline__ void load(const NoblankParams &p, int b, int t0, int tmax)
    {
        const int lane = lane_id();
        const int tlast = tmax > 0 ? tmax - 1 : 0;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const float *row = row_ptr(p, t0 + r < tlast ? t0 + r : tlast, b);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int c = lane + 64 * j;
                v[r][j] = row[c < p.C ? c : p.C - 1];
            }
        }
    }

    // LogSoftmax statistics + emission gather for rows < Tb (NoBlankCTC.py:136, :96-102)
    __device__ __forceinline__ void emit(const NoblankParams &p, const NoblankSmem &sm, int t0, int Tb, int L)
    {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int t = t0 + r;
            if (t >= Tb) break;                              // wave-uniform
            const float ninf = -__builtin_inff();
            float m = ninf;
#pragma unroll
            for (int j = 0; j < CH; ++j) m = fmaxf(m, lane + 64 * j < p.C ? v[r][j] : ninf);
            m = wave_max(m);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) s += lane + 64 * j < p.C ? fast_exp(v[r][j] - m) : 0.f;
            s = wave_sum(s);
            const float lsum = fast_log(s);
            if (lane == 0) { sm.mx[t] = m; sm.ls[t] = lsum; }
            for (int lb = 0; lb < p.SP; lb += kWave) {       // wave-uniform trips (one when S <= 64):
                const int l = lb + lane;                      // every lane must stay active, it is a
                const int k = l < p.SP ? sm.lab[l] : 0;       // bpermute SOURCE for the other lanes
                float xv = 0.f;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float q = __shfl(v[r][j], k & 63, kWave);
                    if ((k >> 6) == j) xv = q;
                }
                if (l < p.SP) sm.em[t * p.SP + l] = (l < L) ? (xv - m) - lsum : kNeg;
            }
        }
    }

    __device__ __forceinline__ void grad(const NoblankParams &p, const NoblankSmem &sm, int b, int t0,
                                         int Tlive, const int (&first)[CH])
    {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            const int t = t0 + r;
            if (t >= p.T) break;                             // wave-uniform
            float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
            if (t < Tlive) {
                const float m = sm.mx[t], lsum = sm.ls[t];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    if (c < p.C) {
                        const float occ = first[j] >= 0 ? sm.be[t * p.SP + first[j]] : 0.f;
                        stream_store(&g[c], p.grad_scale * (fast_exp((v[r][j] - m) - lsum) - occ));
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int c = lane + 64 * j;
                    if (c < p.C) stream_store(&g[c], 0.f);
                }
            }
        }
    }
};

// ---- generic rows (C > 256): strided passes, rows come back from L1/L2 -------------
__device__ __forceinline__ void rows_emit_generic(const NoblankParams &p, const NoblankSmem &sm, int b, int Tb, int L)
{
    const int lane = lane_id();
    for (int t = wave_id(); t < Tb; t += kWaves) {
        const float *row = row_ptr(p, t, b);
        float m = -__builtin_inff();
        for (int c = lane; c < p.C; c += kWave) m = fmaxf(m, row[c]);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < p.C; c += kWave) s += fast_exp(row[c] - m);
        s = wave_sum(s);
        const float lsum = fast_log(s);
        if (lane == 0) { sm.mx[t] = m; sm.ls[t] = lsum; }
        for (int l = lane; l < p.SP; l += kWave)
            sm.em[t * p.SP + l] = (l < L) ? (row[sm.lab[l]] - m) - lsum : kNeg;
    }
}

__device__ __forceinline__ void rows_grad_generic(const NoblankParams &p, const NoblankSmem &sm, int b, int Tlive, int L)
{
    const int lane = lane_id();
    const int G = posterior_group(p.SP), per = kWave / G, sub = lane / G;
    for (int t0 = wave_id() * per; t0 < Tlive; t0 += kWaves * per)
        posterior_row<true>(sm.al, sm.be, sm.em, sm.nxt, sm.dup, t0 + sub, t0 + sub < Tlive, L, p.SP, G);
    __syncthreads();
    for (int t = wave_id(); t < p.T; t += kWaves) {
        const float *row = row_ptr(p, t, b);
        float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
        if (t < Tlive) {
            const float m = sm.mx[t], lsum = sm.ls[t];
            for (int c = lane; c < p.C; c += kWave) {
                const int f = sm.inv[c];
                stream_store(&g[c], p.grad_scale * (fast_exp((row[c] - m) - lsum) - (f >= 0 ? sm.be[t * p.SP + f] : 0.f)));
            }
        } else {
            for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], 0.f);
        }
    }
}

// label tables: lab[] (0 beyond L: those lanes only feed masked cells), first-occurrence
// map inv[], next-duplicate chain nxt[], dup[] flags.  inv/nxt/dup are consumed in P3,
// behind later barriers.
__device__ __forceinline__ void build_label_tables(const NoblankParams &p, const NoblankSmem &sm, int raw_label, int L)
{
    const int tid = threadIdx.x;
    if (tid < p.SP) {                                        // SP <= 256 < kThreads
        int k = 0;
        if (tid < L) {
            k = raw_label % p.C;
            if (k < 0) k += p.C;                             // python negative index (:102)
        }
        sm.lab[tid] = k;
    }
    for (int c = tid; c < p.C; c += kThreads) sm.inv[c] = 0x7fffffff;
    __syncthreads();
    if (tid < L) {
        const int k = sm.lab[tid];
        atomicMin(&sm.inv[k], tid);
        int n = -1;
        for (int l2 = tid + 1; l2 < L; ++l2)
            if (sm.lab[l2] == k) { n = l2; break; }
        sm.nxt[tid] = n;
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += kThreads)
        if (sm.inv[c] == 0x7fffffff) sm.inv[c] = -1;
    if (tid < p.SP) sm.dup[tid] = (tid < L && sm.inv[sm.lab[tid]] == tid && sm.nxt[tid] >= 0) ? 1 : 0;
}

// CH = 0: generic rows (C > 256).  GLB: lattice in global memory (T x S beyond LDS; slower:
// the chains then prefetch their rows from L2) -- completeness path for long sequences.
template <int K, int CH, bool GLB = false>
__global__ __launch_bounds__(kThreads) void noblank_fused_kernel(NoblankParams p)
{
    extern __shared__ float4 smem_raw[];
    const int b = xcd_sample(blockIdx.x, p.B), tid = threadIdx.x, w = wave_id();
    if (tid == kThreads - 64) note_arrival(p.counter, b);       // (the last wave: its first wait is the workgroup barrier)
    const NoblankSmem sm(reinterpret_cast<float *>(smem_raw), p.T, p.SP, p.C,
                         GLB ? p.lattice + (int64_t)b * p.slab : nullptr);
    constexpr int CHR = CH > 0 ? CH : 1;
    Rows<CHR> rows;

    // loads first, oldest = needed first (vmcnt retires in order): the two lengths, this
    // thread's label, then the wave's rows of x -- the label tables are built while the
    // rows are still in flight
    stamp(p, 0);
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const int raw_label = tid < p.S ? load_label(p.lab, p.lab64, (int64_t)b * p.S + tid) : 0;
    if (CH > 0) rows.load(p, b, w * kRows, p.T);

    // contract: 1 <= L_b <= S, L_b <= T_b <= T (the dataset guarantees it,
    // charades_ctc_next_pred.py:609-610); anything else has no alignment.
    const bool ok = L64 >= 1 && L64 <= p.S && Tb64 >= L64 && Tb64 <= p.T;
    const int Tb = ok ? (int)Tb64 : 0, L = ok ? (int)L64 : 0;

    build_label_tables(p, sm, raw_label, L);
    if (tid < 8) sm.dummy[tid] = 0.f;
    for (int i = tid; i < kPrefetch * p.SP; i += kThreads) {          // pad rows of em
        sm.em[i - kPrefetch * p.SP] = kNeg;
        sm.em[p.T * p.SP + i] = kNeg;
    }
    if (CTC_DIAG(p) == 1) return;
    stamp(p, 1);

    // P1 (passes beyond the first are processed first so that pass 0 stays resident)
    if (CH > 0) {
        for (int base = ((Tb - 1) / kRowsPerPass) * kRowsPerPass; base > 0; base -= kRowsPerPass) {
            Rows<CHR> extra;
            extra.load(p, b, base + w * kRows, Tb);
            extra.emit(p, sm, base + w * kRows, Tb, L);
        }
        rows.emit(p, sm, w * kRows, Tb, L);
    } else {
        rows_emit_generic(p, sm, b, Tb, L);
    }
    stamp(p, 2);
    __syncthreads();
    if (CTC_DIAG(p) == 2) return;
    stamp(p, 3);

    // P2
    if (Tb > 0) {
        const bool rot = p.SP <= 63 * K;
        if (w == 0) {
            if (rot) lattice_chain<K, true, true>(sm.em, sm.al, sm.dummy, Tb, L, p.SP);
            else lattice_chain<K, true, false>(sm.em, sm.al, sm.dummy, Tb, L, p.SP);
        } else if (w == 1 && (p.grad || p.gamma)) {
            if (rot) lattice_chain<K, false, true>(sm.em, sm.be, sm.dummy, Tb, L, p.SP);
            else lattice_chain<K, false, false>(sm.em, sm.be, sm.dummy, Tb, L, p.SP);
        }
    }
    stamp(p, 4);
    __syncthreads();
    if (CTC_DIAG(p) == 3) return;
    stamp(p, 5);

    const float nll = ok ? -sm.al[(Tb - 1) * p.SP + (L - 1)] : -kNeg;   // readout, :58-68,139
    // published by the LAST wave: it owns the fewest rows (none when T <= 150), so the
    // store-drain + ticket round trip stays off the other waves' critical path
    if (w == kWaves - 1)
        publish_and_reduce_sum(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter);
    if (!p.grad && !p.gamma) return;

    // P3
    const bool feasible = ok && nll < kInfeasible;
    const int Tlive = feasible ? Tb : 0;
    if (p.gamma) {                                           // posteriors output (no folding): own pass
        const int G = posterior_group(p.SP), per = kWave / G, sub = lane_id() / G;
        for (int t0 = w * per; t0 < p.T; t0 += kWaves * per) {
            const int t = t0 + sub;
            posterior_row<false>(sm.al, sm.be, sm.em, nullptr, nullptr, t, t < Tlive, L, p.SP, G);
            if (t < p.T)
                for (int l = lane_id() % G; l < p.S; l += G)
                    p.gamma[((int64_t)b * p.T + t) * p.S + l] = t < Tlive ? sm.be[t * p.SP + l] : 0.f;
        }
        return;
    }
    if (CH > 0) {
        const int lane = lane_id();
        int first[CHR];
#pragma unroll
        for (int j = 0; j < CHR; ++j) {
            const int c = lane + 64 * j;
            first[j] = (c < p.C) ? sm.inv[c] : -1;
        }
        const int G = posterior_group(p.SP), per = kWave / G, sub = lane / G;
        for (int base = 0; base < p.T; base += kRowsPerPass) {
            const int t0 = base + w * kRows;
            for (int r = 0; r < kRows; r += per)             // this wave's own rows: wave-local
                if (t0 + r < Tlive)
                    posterior_row<true>(sm.al, sm.be, sm.em, sm.nxt, sm.dup, t0 + r + sub,
                                  r + sub < kRows && t0 + r + sub < Tlive, L, p.SP, G);
            if (CTC_DIAG(p) == 4) continue;
            stamp(p, 6);
            if (base == 0) {
                rows.grad(p, sm, b, t0, Tlive, first);
            } else {
                Rows<CHR> extra;
                extra.load(p, b, t0, Tlive);
                extra.grad(p, sm, b, t0, Tlive, first);
            }
        }
    } else {
        rows_grad_generic(p, sm, b, Tlive, L);
    }
    stamp(p, 7);
}

}  // namespace ctc

#include "noblank_pipe.hpp"
#include "noblank_xr.hpp"
#include "noblank_r16.hpp"
#ifdef CTC_AMD_DIAGNOSTICS                                    // measured and not taken: A/B through CTC_AMD_KM=1
#include "noblank_km.hpp"
#endif

namespace ctc {

__global__ __launch_bounds__(256) void scale_grad_kernel(float *g, const float *go, size_t n)
{
    const float s = *go;
    if (s == 1.0f) return;                                   // loss.backward(): nothing to do
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const size_t n4 = n / 4;
        for (size_t k = i; k < n4; k += stride) {
            float4 v = g4[k];
            v.x *= s; v.y *= s; v.z *= s; v.w *= s;
            g4[k] = v;
        }
        for (size_t k = n4 * 4 + i; k < n; k += stride) g[k] *= s;
    } else {
        for (size_t k = i; k < n; k += stride) g[k] *= s;
    }
}

template <int K>
static int launch_noblank(int ch, size_t smem, hipStream_t s, const NoblankParams &p)
{
    const dim3 grid(p.B), block(kThreads);
    if (p.lattice) return launch<noblank_fused_kernel<K, 0, true>>(grid, block, smem, s, p);
    switch (ch) {
        case 1: return launch<noblank_fused_kernel<K, 1>>(grid, block, smem, s, p);
        case 2: return launch<noblank_fused_kernel<K, 2>>(grid, block, smem, s, p);
        case 3: return launch<noblank_fused_kernel<K, 3>>(grid, block, smem, s, p);
        case 4: return launch<noblank_fused_kernel<K, 4>>(grid, block, smem, s, p);
        default: return launch<noblank_fused_kernel<K, 0>>(grid, block, smem, s, p);
    }
}

}  // namespace ctc

using namespace ctc;

// label_smoothing < 0: the plain loss (every kernel); in [0, 1]: the smoothed emission, r16 kernel only
static int noblank_run(const float *x, int64_t stride_t, int64_t stride_b,
                       const void *labels, int labels_i64,
                       const int64_t *in_len, const int64_t *tgt_len,
                       int T, int B, int C, int S,
                       float loss_scale, float grad_scale,
                       float *nll, float *loss, float *grad,
                       void *workspace, void *stream, float label_smoothing)
{
    if (!x || !labels || !in_len || !tgt_len || !nll || !loss || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    int K = 1;
    while (K <= 4 && S > kWave * K) K *= 2;
    if (K > 4) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;         // S <= 256
    NoblankParams p;
    p.x = x; p.st = stride_t; p.sb = stride_b;
    p.lab = labels; p.lab64 = labels_i64;
    p.in_len = in_len; p.tgt_len = tgt_len;
    p.T = T; p.B = B; p.C = C; p.S = S;
    p.SP = (S + K - 1) / K * K;
    static const int debug_stop = diag_env("CTC_AMD_DEBUG_STOP");
    p.stop = debug_stop;
    p.loss_scale = loss_scale; p.grad_scale = grad_scale;
    p.nll = nll; p.loss = loss; p.grad = grad; p.gamma = nullptr;
    static const bool debug_nograd = diag_env("CTC_AMD_DEBUG_NOGRAD") != 0;       // diagnostic: forward only
    if (debug_nograd) p.grad = nullptr;
    p.counter = static_cast<unsigned *>(workspace);
    p.lattice = nullptr; p.slab = 0; p.next_round = 0; p.koff = 0;
    const bool smooth = label_smoothing >= 0.f;
    p.ls_b = smooth ? (1.f - label_smoothing) / (float)C : 0.f;
    p.ls_a = smooth ? label_smoothing - p.ls_b : 1.f;
    size_t smem = noblank_smem_bytes(T, p.SP, C);
    if (smem > kMaxLds) {                                    // long sequence: lattice in the workspace
        if (smooth) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
        smem = noblank_tables_bytes(p.SP, C);
        if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
        p.lattice = reinterpret_cast<float *>(static_cast<char *>(workspace) + 256 + acc_list_bytes(B));
        p.slab = (int64_t)noblank_lattice_floats(T, p.SP);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ch = C <= 256 ? (C + kWave - 1) / kWave : 0;
    // common case (S <= 64, C <= 256, T <= 168): the pipelined schedule
    static const bool no_pipe = diag_env("CTC_AMD_NOPIPE") != 0;
    if (K == 1 && ch >= 1 && T <= kPipeMaxT && !no_pipe && !p.lattice) {
        const dim3 grid(B), block(kThreads);
        const int cus = device_cus();                           // (per device: a process may drive several)
        // extended-range linear lattice (noblank_xr.hpp) wherever its 8-byte cells fit
        static const bool no_xr = diag_env("CTC_AMD_NOXR") != 0;
        const size_t xsmem = xr_smem_bytes(T, p.SP, C);
        const bool dual = B > cus && 2 * smem <= kMaxLds;    // more samples than CUs: two workgroups per CU
        // lean four-rows-per-wave workers (noblank_r16.hpp): 8-byte aligned rows, S <= 31
        static const bool no_r16 = diag_env("CTC_AMD_NOR16") != 0;
        const bool aligned = C % 2 == 0 && stride_t % 2 == 0 && stride_b % 2 == 0 &&
                             reinterpret_cast<uintptr_t>(x) % 8 == 0 && reinterpret_cast<uintptr_t>(grad) % 8 == 0;
        const size_t rsmem = r16_smem_bytes(T, p.SP, C);
        // lean four-rows-per-wave workers (noblank_r16.hpp): 8-byte aligned rows, S <= 31.  Also with more
        // samples than CUs: one workgroup per CU at a time still beats two of the one-row-per-wave kind
        // (T = 150, C = 158, us per launch r16 / xr: B = 384 22.5 / 27.7, 512 25.9 / 31.2, 768 37.6 / 45.2,
        // 1024 49.7 / 56.6, 1536 78.2 / 78.5, 2048 101.3 / 99.7)
        int n4 = 0, n2 = 0;
        if (!no_r16 && aligned && r16_shape(C, n4, n2) && p.SP <= 31 && rsmem <= kMaxLds) {
            // logits + gradient beyond the memory-side cache (256 MB): non-temporal gradient stores
            const bool nt = (size_t)8 * T * B * C > ((size_t)230 << 20);
#ifdef CTC_AMD_DIAGNOSTICS
            // the chains split over exponent and mantissa waves (noblank_km.hpp) wherever its arrays fit and the
            // number of lattice paths leaves two mantissas room in fp32
            static const bool use_km = diag_env("CTC_AMD_KM") != 0;
            const size_t ksmem = km_smem_bytes(T, p.SP, C);
            const int koff = km_offset(T, S);
            if (use_km && ksmem <= kMaxLds && koff >= 0) {
                p.koff = koff;
                p.next_round = (B > cus && (size_t)T * ((C * 4 + 127) / 128) <= (size_t)kKmWorkers * kWave) ? cus : 0;
#define CTC_KM_CASE(K, A, Bq)                                                                           \
                case K: return nt ? launch<noblank_km_kernel<A, Bq, true>>(grid, block, ksmem, s, p)      \
                                  : launch<noblank_km_kernel<A, Bq, false>>(grid, block, ksmem, s, p);
                switch (4 * n4 + n2) {
                    CTC_KM_CASE(1, 0, 1) CTC_KM_CASE(2, 0, 2) CTC_KM_CASE(4, 1, 0) CTC_KM_CASE(5, 1, 1)
                    CTC_KM_CASE(6, 1, 2) CTC_KM_CASE(8, 2, 0) CTC_KM_CASE(9, 2, 1) CTC_KM_CASE(10, 2, 2)
                    CTC_KM_CASE(12, 3, 0) CTC_KM_CASE(13, 3, 1) CTC_KM_CASE(14, 3, 2)
                    default: return nt ? launch<noblank_km_kernel<4, 0, true>>(grid, block, ksmem, s, p)
                                       : launch<noblank_km_kernel<4, 0, false>>(grid, block, ksmem, s, p);
                }
#undef CTC_KM_CASE
            }
#endif
            p.next_round = (B > cus && (size_t)T * ((C * 4 + 127) / 128) <= (size_t)kPipeWorkers * kWave) ? cus : 0;
            // More than two rounds of samples: one PERSISTENT workgroup per CU that keeps the next sample's rows in a
            // second set of registers (noblank_r16.hpp), wherever two sets fit the 128 VGPRs of a 16-wave workgroup
            // (C <= 192) and a gradient is wanted.  T = 150, C = 158, us per launch persistent / one sample per
            // workgroup: B = 2048 82.7 / 98.0, 1024 41.7 / 45.1, 512 24.3 / 24.1.
            static const bool no_ps = diag_env("CTC_AMD_NOPS") != 0;
            if (B > 2 * cus && cus > 0 && !no_ps && p.grad && 4 * n4 + 2 * n2 <= 12) {
                const dim3 pgrid(cus);
#define CTC_R16_CASE(K, A, Bq)                                                                          \
                case K: return nt ? launch<noblank_r16_kernel<A, Bq, true, true>>(pgrid, block, rsmem, s, p)   \
                                  : launch<noblank_r16_kernel<A, Bq, false, true>>(pgrid, block, rsmem, s, p);
                switch (4 * n4 + n2) {
                    CTC_R16_CASE(1, 0, 1) CTC_R16_CASE(2, 0, 2) CTC_R16_CASE(4, 1, 0) CTC_R16_CASE(5, 1, 1)
                    CTC_R16_CASE(6, 1, 2) CTC_R16_CASE(8, 2, 0) CTC_R16_CASE(9, 2, 1) CTC_R16_CASE(10, 2, 2)
                    CTC_R16_CASE(12, 3, 0)
                    default: break;
                }
#undef CTC_R16_CASE
            }
#define CTC_R16_CASE(K, A, Bq)                                                                          \
            case K: return nt ? launch<noblank_r16_kernel<A, Bq, true>>(grid, block, rsmem, s, p)         \
                              : launch<noblank_r16_kernel<A, Bq, false>>(grid, block, rsmem, s, p);
            switch (4 * n4 + n2) {
                CTC_R16_CASE(1, 0, 1) CTC_R16_CASE(2, 0, 2) CTC_R16_CASE(4, 1, 0) CTC_R16_CASE(5, 1, 1)
                CTC_R16_CASE(6, 1, 2) CTC_R16_CASE(8, 2, 0) CTC_R16_CASE(9, 2, 1) CTC_R16_CASE(10, 2, 2)
                CTC_R16_CASE(12, 3, 0) CTC_R16_CASE(13, 3, 1) CTC_R16_CASE(14, 3, 2)
                default: return nt ? launch<noblank_r16_kernel<4, 0, true>>(grid, block, rsmem, s, p)
                                   : launch<noblank_r16_kernel<4, 0, false>>(grid, block, rsmem, s, p);
            }
#undef CTC_R16_CASE
        }
        if (smooth) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;    // the smoothed emission lives in the kernel above only
        if (!no_xr && xsmem <= kMaxLds && (!dual || 2 * xsmem <= kMaxLds)) {
            if (dual) {
                switch (ch) {
                    case 1: return launch<noblank_xr_kernel<1, true>>(grid, block, xsmem, s, p);
                    case 2: return launch<noblank_xr_kernel<2, true>>(grid, block, xsmem, s, p);
                    case 3: return launch<noblank_xr_kernel<3, true>>(grid, block, xsmem, s, p);
                    default: return launch<noblank_xr_kernel<4, true>>(grid, block, xsmem, s, p);
                }
            }
            switch (ch) {
                case 1: return launch<noblank_xr_kernel<1, false>>(grid, block, xsmem, s, p);
                case 2: return launch<noblank_xr_kernel<2, false>>(grid, block, xsmem, s, p);
                case 3: return launch<noblank_xr_kernel<3, false>>(grid, block, xsmem, s, p);
                default: return launch<noblank_xr_kernel<4, false>>(grid, block, xsmem, s, p);
            }
        }
        if (dual) {
            switch (ch) {
                case 1: return launch<noblank_pipelined_kernel<1, true>>(grid, block, smem, s, p);
                case 2: return launch<noblank_pipelined_kernel<2, true>>(grid, block, smem, s, p);
                case 3: return launch<noblank_pipelined_kernel<3, true>>(grid, block, smem, s, p);
                default: return launch<noblank_pipelined_kernel<4, true>>(grid, block, smem, s, p);
            }
        }
        switch (ch) {
            case 1: return launch<noblank_pipelined_kernel<1, false>>(grid, block, smem, s, p);
            case 2: return launch<noblank_pipelined_kernel<2, false>>(grid, block, smem, s, p);
            case 3: return launch<noblank_pipelined_kernel<3, false>>(grid, block, smem, s, p);
            default: return launch<noblank_pipelined_kernel<4, false>>(grid, block, smem, s, p);
        }
    }
    if (smooth) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    switch (K) {
        case 1: return launch_noblank<1>(ch, smem, s, p);
        case 2: return launch_noblank<2>(ch, smem, s, p);
        default: return launch_noblank<4>(ch, smem, s, p);
    }
}

extern "C" int ctc_amd_noblank_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                                         const void *labels, int labels_i64,
                                         const int64_t *in_len, const int64_t *tgt_len,
                                         int T, int B, int C, int S,
                                         float loss_scale, float grad_scale,
                                         float *nll, float *loss, float *grad,
                                         void *workspace, void *stream)
{
    return noblank_run(x, stride_t, stride_b, labels, labels_i64, in_len, tgt_len, T, B, C, S, loss_scale, grad_scale,
                       nll, loss, grad, workspace, stream, -1.f);
}

extern "C" int ctc_amd_noblank_smoothed_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                                                  const void *labels, int labels_i64,
                                                  const int64_t *in_len, const int64_t *tgt_len,
                                                  int T, int B, int C, int S, float label_smoothing,
                                                  float loss_scale, float grad_scale,
                                                  float *nll, float *loss, float *grad,
                                                  void *workspace, void *stream)
{
    if (!(label_smoothing >= 0.f && label_smoothing <= 1.f)) return CTC_AMD_ERR_BAD_ARGUMENT;
    return noblank_run(x, stride_t, stride_b, labels, labels_i64, in_len, tgt_len, T, B, C, S, loss_scale, grad_scale,
                       nll, loss, grad, workspace, stream, label_smoothing);
}

extern "C" int ctc_amd_scale_grad(float *grad, const float *grad_out, size_t n, void *stream)
{
    if (!grad || !grad_out) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (n == 0) return 0;
    // in the common case (grad_out == 1) every block only reads one float and leaves: the launch costs one kernel boundary,
    // 1.7 us in a graph, for any grid up to 2048 blocks (tools/scale_grad_time.py: 64 blocks 1.68, 1024 blocks 1.72 us).
    // When the gradient must be scaled -- `(loss / accum_steps).backward()`, a loss scaler -- the grid is what moves the
    // 2 x 24 MB of config 2: 64 blocks 17.7 us, 256 blocks 7.8, 1024 blocks 5.9 (round 3; it was 64).
    size_t blocks = (n / 4 + 255) / 256;
    static const int cap = diag_env("CTC_AMD_SCALE_BLOCKS", 1024);
    if (blocks > (size_t)cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_grad_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), grad, grad_out, n);
    return (int)hipGetLastError();
}

extern "C" int ctc_amd_noblank_posteriors(const float *x, int64_t stride_t, int64_t stride_b,
                                          const void *labels, int labels_i64,
                                          const int64_t *in_len, const int64_t *tgt_len,
                                          int T, int B, int C, int S,
                                          float *nll, float *gamma,
                                          void *workspace, void *stream)
{
    if (!x || !labels || !in_len || !tgt_len || !nll || !gamma || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    int K = 1;
    while (K <= 4 && S > kWave * K) K *= 2;
    if (K > 4) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    NoblankParams p;
    p.x = x; p.st = stride_t; p.sb = stride_b;
    p.lab = labels; p.lab64 = labels_i64;
    p.in_len = in_len; p.tgt_len = tgt_len;
    p.T = T; p.B = B; p.C = C; p.S = S;
    p.SP = (S + K - 1) / K * K;
    p.stop = 0;
    p.loss_scale = 0.f; p.grad_scale = 0.f;
    p.nll = nll; p.grad = nullptr; p.gamma = gamma;
    // the batch-mean slot of the in-launch reduction lands in a spare workspace word
    p.counter = static_cast<unsigned *>(workspace);
    p.loss = reinterpret_cast<float *>(static_cast<char *>(workspace) + 32);
    p.lattice = nullptr; p.slab = 0; p.next_round = 0; p.ls_a = 1.f; p.ls_b = 0.f; p.koff = 0;
    size_t smem = noblank_smem_bytes(T, p.SP, C);
    if (smem > kMaxLds) {
        smem = noblank_tables_bytes(p.SP, C);
        if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
        p.lattice = reinterpret_cast<float *>(static_cast<char *>(workspace) + 256 + acc_list_bytes(B));
        p.slab = (int64_t)noblank_lattice_floats(T, p.SP);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ch = C <= 256 ? (C + kWave - 1) / kWave : 0;    // phase-serial kernel: any supported shape
    {   // the four-rows-per-wave kernel of the loss (noblank_r16.hpp) wherever it takes the shape: same chains, the
        // gradient phase cut off after the row-normalised posteriors
        int n4 = 0, n2 = 0;
        const bool aligned = C % 2 == 0 && stride_t % 2 == 0 && stride_b % 2 == 0 && reinterpret_cast<uintptr_t>(x) % 8 == 0;
        const size_t rsmem = r16_smem_bytes(T, p.SP, C);
        if (K == 1 && T <= kPipeMaxT && !p.lattice && aligned && r16_shape(C, n4, n2) && p.SP <= 31 && rsmem <= kMaxLds) {
            const dim3 grid(B), block(kThreads);
#define CTC_R16_CASE(K, A, Bq) case K: return launch<noblank_r16_kernel<A, Bq, false>>(grid, block, rsmem, s, p);
            switch (4 * n4 + n2) {
                CTC_R16_CASE(1, 0, 1) CTC_R16_CASE(2, 0, 2) CTC_R16_CASE(4, 1, 0) CTC_R16_CASE(5, 1, 1)
                CTC_R16_CASE(6, 1, 2) CTC_R16_CASE(8, 2, 0) CTC_R16_CASE(9, 2, 1) CTC_R16_CASE(10, 2, 2)
                CTC_R16_CASE(12, 3, 0) CTC_R16_CASE(13, 3, 1) CTC_R16_CASE(14, 3, 2)
                default: return launch<noblank_r16_kernel<4, 0, false>>(grid, block, rsmem, s, p);
            }
#undef CTC_R16_CASE
        }
    }
    switch (K) {
        case 1: return launch_noblank<1>(ch, smem, s, p);
        case 2: return launch_noblank<2>(ch, smem, s, p);
        default: return launch_noblank<4>(ch, smem, s, p);
    }
}

#ifdef CTC_AMD_DIAGNOSTICS
// Diagnostic entry point (tools/chain_probe.py), not part of include/ctc_amd.h: the lattice chains of
// noblank_r16.hpp alone, every emission row pre-published; out[0], out[1] = shader cycles of the
// alpha / beta chain.
extern "C" int ctc_amd_debug_chain_probe(int T, int SP, int waves_alive, int grid, void *out, void *stream, int mode,
                                         int chain_b)
{
    using namespace ctc;
    NoblankParams p = {};
    p.T = T; p.SP = SP; p.S = SP; p.B = grid; p.C = 32;
    p.stop = -77;                                            // the chains also time their main loop: out[8], out[9]
    p.counter = reinterpret_cast<unsigned *>(static_cast<unsigned long long *>(out) + 8);
    const size_t smem = r16_smem_bytes(T, SP, 158);
    return launch<r16_chain_probe_kernel>(dim3(grid), dim3(kThreads), smem, static_cast<hipStream_t>(stream), p,
                                          static_cast<unsigned long long *>(out), waves_alive, mode, chain_b);
}
#endif
