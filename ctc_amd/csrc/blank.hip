// Standard blank-CTC loss + gradient for gfx950 (BASELINE config 5: long sequences).
//
// Semantics: torch.nn.CTCLoss(blank, reduction='mean', zero_infinity=False) as the
// reference uses it at models/layers/AsyncTFCriterion.py:198,319-321 (arithmetic lives in
// aten::_ctc_loss, not in the reference repo): extended label l' of n = 2L+1 states,
//   alpha_t(s) = LSE(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != blank
//                and l'_s != l'_{s-2}]) + lp_t(l'_s),      nll = -LSE(alpha_{T-1}(n-1), alpha_{T-1}(n-2)),
//   grad[t,c] = (exp(lp[t,c]) - sum_{s: l'_s = c} gamma_t(s)) * grad_scale / max(L,1), 0 for t >= T_b.
//
// T x (2S+1) does not fit in LDS (config 5: 2000 x 201), so the lattice lives in the
// caller's workspace.  Three kinds of work with different shapes:
//   gather  (bandwidth)   em[b,t,s] = lp[t,b,l'_s]: rows of lp are read once, in whole lines
//   chains  (latency)     one workgroup per sample, wave 0 = alpha, wave 1 = beta, K states per
//                         lane, the two neighbour states through DPP wave shifts, emission rows
//                         prefetched 64/K deep; T dependent steps of ~150 ns each
//   grad    (bandwidth)   one wave per (t,b) row: gamma_t = softmax_s(alpha+beta-e)
//                         (row-normalised, see lattice.hpp), folded per class (blank by a wave
//                         reduction, repeated labels by the first-occurrence chain), then the
//                         dense row exp(lp) - occupancy with coalesced loads / stores.
//
// Two schedules:
//  * three launches (blank_gather_kernel, blank_chain_kernel, blank_grad_kernel), the chains
//    alone on 2 B waves while 250 CUs idle -- forward-only calls, and every shape where the
//    second schedule does not pay (run_blank has the measurements);
//  * ONE persistent launch (blank_fused_kernel, after the tiny blank_tables_kernel) of one
//    512-thread workgroup per CU.  Workgroups [0, B) own one sample each: waves 0 / 1 run the
//    alpha / beta chains and touch nothing but LDS and their write-only lattice rows; six loader
//    waves (three per direction) gather the emission rows straight from log_probs, dozens of
//    rows ahead, into an LDS ring per direction -- the emissions never go to HBM, and no load
//    latency sits on the chains (beta rows are stored WITHOUT their own emission, so
//    gamma = alpha + beta' needs no emission either).  All other workgroups are row workers:
//    they turn a row into gradient as soon as BOTH chains have passed it (the chains cross in
//    the middle of the sample, so rows become ready from the middle outwards, two per step),
//    in that order.  Hand-off chains -> workers: `steps landed` per sample and direction in the
//    workspace.  Forward progress needs every workgroup resident at once: the grid is at most
//    one workgroup per CU (the LDS request makes that the occupancy too), and chain workgroups
//    have the lowest block indices, so they are placed first.  All waits are bounded
//    (about a second); a wait that runs out raises the status word and poisons its outputs with NaN.
//    Cross-XCD visibility (each XCD has its own L2): lattice rows are written with agent-scope
//    (sc1, write-through) stores and read with sc1 loads; the counters are agent-scope atomics.
#include <cstdlib>

#include "common.hpp"
#include <atomic>
#include <cstdlib>

#include "launch.hpp"

namespace ctc {

constexpr float kNegB = -1.0e30f;    // finite stand-in for -inf inside the scans
constexpr int kRingRegs = 64;        // three-launch chains: VGPRs of emission rows in flight (64/K rows;
                                     // 8 rows stalled the chain on HBM latency, 16 rows gained 7 %)
constexpr int kSpinLimit = 1 << 18;  // looks at a chain's progress (>= 3 us apart: ~1 s) before a worker gives up
constexpr int kLdsSpinLimit = 1 << 23;   // polls of an LDS flag (~0.1 us apart: ~1 s) before a chain / loader gives up
constexpr int kSyncHead = 64;        // ints in front of the counters (status word)
constexpr int kProgPitch = 32;       // ints between two progress counters: one 128-byte line each (they are polled)
constexpr int kAuxAgent = 16;        // buffer-instruction cache policy: sc1 = agent scope
constexpr int kFusedMinT = 128;      // the persistent launch needs at least this many steps (and pays from twice as many)
constexpr int kMaxV4 = 4;            // float4 per lane that cover a row of the VEC4 paths (C <= 1024)
constexpr int kFusedWaves = 8;       // waves of a workgroup of the fused launch
constexpr int kFusedThreads = kFusedWaves * kWave;
constexpr int kRingRows = 128;       // x 1/K: emission rows per direction in the LDS ring (32 KB for every K)
constexpr int kLoadAhead = 12;       // rows each loader wave keeps in flight
constexpr int kLoaders = 3;          // loader waves per direction (two could not keep up: the chains waited 30 % of the time)
constexpr int kGroup = 16;           // chain steps between hand-off checks (capped at half a ring; 8: +2 %, 4: +6 %)
constexpr int kLandLag = 16;         // lattice stores that may still be in flight when progress is published (0..24
                                     // measure the same, 48 is 1.5 % slower: the workers hear of rows later; with
                                     // half of the lattice in memory 16 steps = 8 stores: 32 steps is 0.8 % slower)
constexpr size_t kFusedLdsHead = 64; // bytes of LDS flags in front of the rings
constexpr int kPastLattice = 1 << 30; // a byte offset beyond any sample's lattice (out-of-range buffer accesses are dropped)
// Persistent launch: only every other lattice row goes to memory -- alpha on even rows, beta' on odd rows -- and the
// worker of a row PAIR (2P, 2P+1) redoes one forward and one backward step from the two log_probs rows it holds
// anyway (0.2 GB less traffic at B=64 T=2000 S=100, and half the chains' stores).
#ifdef CTC_X_FULL_LATTICE
constexpr bool kHalfLattice = false;
#else
constexpr bool kHalfLattice = true;
#endif
// Persistent launch, float4 loaders: every log_probs row is gathered ONCE.  The loaders of a direction gather the first
// half of its steps from log_probs; its CHAIN, which has each emission row in registers anyway, leaves it in the
// workspace (p.em, the three-launch schedule's emission table; plain stores -- producer and consumer are waves of ONE
// workgroup, so the CU's L2 is the meeting point); the loaders of the OTHER direction -- for which these rows are the
// second half -- read the 4 K bytes per lane back instead of whole log_probs rows (0.3 GB less traffic at B=64 T=2000
// C=1000 S=100, and light loaders while the row workers use the memory system).
// Tried on the way: a storer wave per direction (ten waves per workgroup leave 168 VGPRs per wave, the loaders' twelve
// rows in flight spill: 860 us); the loaders storing their own rows (one more line request per row on the busiest
// unit of the phase: first half 50 instead of 41 us per 256 steps); one loader loop with a branch per row (the
// compiler must then assume the fewest memory operations behind a row it waits for: three rows in flight, 730 us).
// (Also measured and removed again: the chains of this launch on (mantissa, exponent) states as in noblank_r16.hpp --
// 52 plain VALU operations per step at four states per lane instead of twelve transcendentals + 40, the loaders
// splitting each emission into (2^frac, floor), lattice rows still stored as log2 values: correct, and 463 against
// 430 us on the same device.  The chain's own time per step did not move (0.155 us: it is not bound by its arithmetic
// but by the per-group hand-offs and the landing window of its stores) and it waited longer for its loaders.)
// MEASURED, NOT ON: 1.94 instead of 2.13 GB and a second half of the chains at 36-46 instead of 44-67 us per 256 steps,
// but the extra store per step sits on the busiest unit of the first half (the sample's CU issues every log_probs
// line request of six loaders): the chains cross at 209 instead of 170 us and the launch takes 459 against 447 us.
#ifdef CTC_X_GATHER_ONCE
constexpr bool kGatherOnce = true;
#else
constexpr bool kGatherOnce = false;
#endif

// Persistent launch, float4 rows: the IDLE WORKER POOL gathers the emission rows.  Until the chains cross in the middle
// of the samples the row workers have nothing to do (38 % of the launch at B=64 T=2000: 192 CUs idle, HBM a third
// used) while each sample's CU issues every log_probs line request of its six loaders, stages the 4-KB rows in LDS and
// gathers them -- and its chain still waits 17 % of that phase for emission rows (34 % later, beside the streaming
// workers: tools/blank_stamps.py, CTC_AMD_BLANK_DEBUG=384).  So the pool does that work first: every log_probs row is
// read ONCE, in the order in which the chains want it (row t of a sample is wanted at distance min(t, T_b-1-t) from
// the nearer end: by alpha at step t, by beta at step T_b-1-t), gathered by class into the 4 K bytes per lane the
// chains consume (log2 units) and left in the workspace (p.em, the three-launch schedule's emission table) with
// write-through stores; then the pool turns to the gradient as before.  A sample's workgroup keeps two light loader
// waves per direction (0.8 KB per row instead of 4 KB, no LDS staging, no gather) and one SCOUT wave per direction.
// Hand-off pool -> sample: rows are counted per (sample, chunk of kPoolChunk distances) with an agent-scope atomic
// AFTER the gathering wave's stores have landed (s_waitcnt vmcnt(0)); the scout polls the counters in order and
// publishes "chunks complete" in LDS; a loader issues the load of a row only when its chunk is known complete (the
// flag is read, and waited for, BEFORE the row is loaded: loads of one wave may be served out of order).  The second
// half of a chain's steps are rows the other direction needed first: no further checks.  Producers never wait.
#ifdef CTC_X_NO_POOL_GATHER
constexpr bool kPoolGather = false;
#else
constexpr bool kPoolGather = true;
#endif
constexpr int kPoolChunk = 16;       // distances per hand-off counter
constexpr int kPoolBatch = 4;        // rows a gathering wave keeps in flight
constexpr int kPoolLoaders = 2;      // data loaders per direction when the pool gathers (the third loader wave is the scout)

__host__ __device__ inline int blank_pool_chunks(int T) { return (((T + 1) >> 1) + kPoolChunk - 1) / kPoolChunk; }
// ints of hand-off state behind the per-sample tables (see BlankParams::sync): status word, two progress counters
// per sample (one 128-byte line each), the pool-gather chunk counters [B][chunks]
int blank_sync_ints(int T, int B)
{
    return kSyncHead + 2 * ((B + 63) & ~63) * kProgPitch + ((B * blank_pool_chunks(T) + 63) & ~63);
}

struct BlankParams {
    const float *lp;
    int64_t st, sb;
    const void *tgt;
    int tgt64;
    const int64_t *in_len, *tgt_len;
    int T, B, C, S, NSP, blank;      // NSP = 64*K padded state count
    float loss_scale, grad_scale;
    float *nll, *loss, *grad;
    unsigned *counter;
    float *em, *al, *be;             // [B][T][NSP] each
    int *cls, *nxt, *first;          // [B][NSP]: class of state s; next state with the same class;
                                     // 1 when s is the first state carrying its (non-blank) class
    int2 *meta;                      // [B] (T_b, or 0 when there is no alignment: T_b < L_b + adjacent repeats; L_b)
    // fused schedule: sync[0] status, sync[kSyncHead + (dir Bp + b) kProgPitch]
    // steps of chain `dir` of sample b whose lattice rows have landed
    int *sync;
    int *chunk;                      // [B][nchunk] rows of that chunk of distances gathered by the pool (fused schedule)
    int Bp, nsync, nchunk;
    int debug;                       // CTC_AMD_BLANK_DEBUG (diagnostics: 128 = timeline stamps)
};

__device__ __forceinline__ bool blank_sample_ok(const BlankParams &p, int b, int &Tb, int &L)
{
    const int64_t Tb64 = p.in_len[b], L64 = p.tgt_len[b];
    const bool ok = L64 >= 0 && L64 <= p.S && Tb64 >= 0 && Tb64 <= p.T;
    Tb = ok ? (int)Tb64 : 0;
    L = ok ? (int)L64 : 0;
    return ok;
}

// ---- hand-off primitives of the fused schedule ------------------------------------------------------
__device__ __forceinline__ int agent_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void agent_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int wg_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void wg_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// true once *flag >= target, a flag in LDS that another wave of the workgroup bumps; false when the
// bounded wait ran out (status word raised).  `seen`: the last value read, so that the caller only
// comes back when it needs more
__device__ __forceinline__ bool lds_wait_ge(const BlankParams &p, const int *flag, int target, int &seen)
{
    bool ok = false;
    for (int it = 0; it < kLdsSpinLimit; ++it) {
        seen = wg_load(flag);
        if (seen >= target) { ok = true; break; }
        if ((it & 4095) == 4095 && agent_load(p.sync) != 0) break;   // somebody already gave up
        __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) { agent_store(p.sync, 1); raise_status(p.counter, kStatusBlankStarved); }
    asm volatile("" ::: "memory");
    return ok;
}

// The same wait without any vector-memory operation inside (no look at the global status word, no status raised here):
// for loops that keep vector loads in flight -- a vector-memory operation on a side path makes the compiler's
// s_waitcnt bookkeeping assume the worst at the join, and "twelve rows in flight" silently become one.  The caller
// raises the status when this returns false.
__device__ __forceinline__ bool lds_wait_ge_quiet(const int *flag, int target, int &seen)
{
    for (int it = 0; it < kLdsSpinLimit; ++it) {
        seen = wg_load(flag);
        if (seen >= target) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// diagnostics (CTC_AMD_BLANK_DEBUG & 128): 100-MHz timestamps into workspace bytes [64,256), tools/blank_stamps.py
__device__ __forceinline__ void bstamp(const BlankParams &p, int slot)
{
    if (!(p.debug & 128) || lane_id() != 0 || slot >= 24) return;
    reinterpret_cast<unsigned long long *>(p.counter)[8 + slot] = __builtin_amdgcn_s_memrealtime();
}

typedef int i4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef int i2_t __attribute__((ext_vector_type(2)));

// one sample's [T][NSP] lattice as a raw buffer (bounds-checked by the hardware)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t lattice_rsrc(const float *base, int T, int NSP)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, T * NSP * (int)sizeof(float), 0x00020000);
}
template <int K, int AUX = kAuxAgent>
__device__ __forceinline__ void agent_store_row(__amdgpu_buffer_rsrc_t r, int byte_off, const float (&a)[K])
{
    if constexpr (K == 2) {
        const i2_t v = {__builtin_bit_cast(int, a[0]), __builtin_bit_cast(int, a[1])};
        __builtin_amdgcn_raw_buffer_store_b64(v, r, byte_off, 0, AUX);
    } else {
#pragma unroll
        for (int q = 0; q < K; q += 4) {
            const i4_t v = {__builtin_bit_cast(int, a[q]), __builtin_bit_cast(int, a[q + 1]),
                            __builtin_bit_cast(int, a[q + 2]), __builtin_bit_cast(int, a[q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off + q * 4, 0, AUX);
        }
    }
}
template <int K, int AUX = kAuxAgent>
__device__ __forceinline__ void agent_load_row(__amdgpu_buffer_rsrc_t r, int byte_off, float (&a)[K])
{
    if constexpr (K == 2) {
        const i2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, AUX);
        const int x = v.x, y = v.y;                          // (element reads of a vector go through temporaries)
        a[0] = __builtin_bit_cast(float, x);
        a[1] = __builtin_bit_cast(float, y);
    } else {
#pragma unroll
        for (int q = 0; q < K; q += 4) {
            const i4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + q * 4, 0, AUX);
            const int x = v.x, y = v.y, z = v.z, w = v.w;
            a[q] = __builtin_bit_cast(float, x);
            a[q + 1] = __builtin_bit_cast(float, y);
            a[q + 2] = __builtin_bit_cast(float, z);
            a[q + 3] = __builtin_bit_cast(float, w);
        }
    }
}

// ---- K0: state tables + emission gather -------------------------------------------------
// classes of the extended label into LDS (all threads of the block; ends with a barrier)
__device__ __forceinline__ void blank_classes(const BlankParams &p, int b, int n, int *s_cls)
{
    for (int s = threadIdx.x; s < p.NSP; s += blockDim.x) {
        int c = p.blank;
        if (s < n && (s & 1)) {
            c = load_label(p.tgt, p.tgt64, (int64_t)b * p.S + (s >> 1));
            c = c < 0 ? 0 : (c >= p.C ? p.C - 1 : c);        // memory safety for bad labels
        }
        s_cls[s] = c;
    }
    __syncthreads();
}

// per-sample state tables + (effective length, target length), from the classes in LDS
__device__ __forceinline__ void blank_tables(const BlankParams &p, int b, int Tb, int L, const int *s_cls)
{
    const int n = 2 * L + 1;
    for (int s = threadIdx.x; s < p.NSP; s += blockDim.x) {
        const int c = s_cls[s];
        p.cls[b * p.NSP + s] = c;
        // one pass over the labels, no early exit: the LDS reads are independent (and the same address for
        // every thread), so they pipeline -- two scans that stop at the first match cost 14 us at L = 100
        const bool label = s < n && (s & 1);
        int nx = -1, fi = label ? 1 : 0;
        if (label) {
#pragma unroll 8
            for (int s2 = 1; s2 < n; s2 += 2) {
                const bool same = s_cls[s2] == c;
                if (same && s2 < s) fi = 0;
                if (same && s2 > s && nx < 0) nx = s2;
            }
        }
        p.nxt[b * p.NSP + s] = nx;
        p.first[b * p.NSP + s] = fi;
    }
    if (threadIdx.x == 0) {                                  // an alignment needs one step per label plus a blank
        int need = L;                                        // between every two equal neighbours
        for (int l = 1; l < L; ++l) need += s_cls[2 * l + 1] == s_cls[2 * l - 1] ? 1 : 0;
        p.meta[b] = make_int2(Tb >= need ? Tb : 0, L);
    }
}

// three-launch schedule.  Grid: (row blocks, B); block x takes rows [x rpb, (x+1) rpb)
__global__ __launch_bounds__(256) void blank_gather_kernel(BlankParams p, int rows_per_block)
{
    extern __shared__ int s_cls[];                           // [NSP]
    const int b = blockIdx.y, tid = threadIdx.x;
    int Tb, L;
    blank_sample_ok(p, b, Tb, L);
    const int n = 2 * L + 1;
    blank_classes(p, b, n, s_cls);
    if (blockIdx.x == 0) blank_tables(p, b, Tb, L, s_cls);
    const int t_begin = blockIdx.x * rows_per_block;
    const int t_end = min(t_begin + rows_per_block, Tb);
    for (int t = t_begin; t < t_end; ++t) {
        const float *row = p.lp + (int64_t)t * p.st + (int64_t)b * p.sb;
        float *out = p.em + ((int64_t)b * p.T + t) * p.NSP;
        for (int s = tid; s < p.NSP; s += blockDim.x) out[s] = s < n ? fmaxf(row[s_cls[s]] * kLog2e, kNegB) : kNegB;
    }
}

// fused schedule: tables only, and the hand-off counters back to zero.  Grid: B
__global__ __launch_bounds__(256) void blank_tables_kernel(BlankParams p)
{
    extern __shared__ int s_cls[];                           // [NSP]
    const int b = blockIdx.x;
    int Tb, L;
    blank_sample_ok(p, b, Tb, L);
    blank_classes(p, b, 2 * L + 1, s_cls);
    blank_tables(p, b, Tb, L, s_cls);
    for (int i = b * blockDim.x + threadIdx.x; i < p.nsync; i += gridDim.x * blockDim.x) p.sync[i] = 0;
}

// ---- K1: alpha / beta chains ----------------------------------------------------------------
// The blank lattice (em, alpha, beta) is kept in LOG2 units: v_exp_f32 / v_log_f32 are base-2,
// so a state update is max, subtract, exp2, add, log2, add with no base-conversion multiplies.
// Only differences alpha+beta-em and the final likelihood (x ln 2) leave the lattice.
__device__ __forceinline__ float lse2_2(float a, float b)
{
    const float m = vmax(a, b);
    return m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}

// per-lane constants of a chain: which of its K states are real, and which take the s-2 edge
template <int K, bool FWD>
__device__ __forceinline__ void blank_state_flags(const BlankParams &p, int b, int n, bool (&skip)[K], bool (&valid)[K])
{
    const int s0 = lane_id() * K;
    const int *cls = p.cls + b * p.NSP;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        valid[k] = s < n;
        // alpha: from s-2 when l'_s is a label differing from l'_{s-2};  beta: from s+2 likewise
        const int s2 = FWD ? s - 2 : s + 2;
        skip[k] = (s & 1) && s2 >= 0 && s2 < n && cls[s] != cls[s2];
    }
}

// one time step of a chain: a <- LSE of the predecessors + e
// A label state's three predecessors are itself and the two predecessors of the BLANK state next to it (alpha: s-1
// and s-2 are what blank s-1 comes from; beta: s+1 and s+2, blank s+1), so its sum is 2^a + 2^(that blank's
// pre-emission value): every state is a two-term log-sum-exp, 8 transcendentals per step at four states per lane
// instead of 12 (they are half of the step's issue time).  Where the s-2 edge does not exist (repeated label) the
// second term is the neighbour itself.  Costs one more fp32 rounding on that path (the blank's value is rounded before
// it is reused); the workers' recomputed rows go through this same function, so they still match the chains' bit for bit.
typedef float v2f_t __attribute__((ext_vector_type(2)));
#ifdef CTC_X_SCALAR_STEP
constexpr bool kPackedStep = false;
#else
constexpr bool kPackedStep = true;
#endif
// two two-term log-sum-exps at once: the subtractions and additions go through the packed fp32 pipe (v_pk_add_f32: two
// lanes' worth per instruction), the maxima and the transcendentals stay single -- 9 instead of 12 VALU operations
__device__ __forceinline__ v2f_t lse2_2x2(v2f_t a, v2f_t b)
{
    const v2f_t d = a - b;
    v2f_t t;
    t.x = __builtin_amdgcn_exp2f(-fabsf(d.x));
    t.y = __builtin_amdgcn_exp2f(-fabsf(d.y));
    t += 1.0f;
    v2f_t l, m;
    l.x = __builtin_amdgcn_logf(t.x);
    l.y = __builtin_amdgcn_logf(t.y);
    m.x = vmax(a.x, b.x);                                    // (fmaxf would canonicalise both inputs first: eight more VALU operations per step)
    m.y = vmax(a.y, b.y);
    return m + l;
}

template <int K, bool FWD>
__device__ __forceinline__ void blank_step(float (&a)[K], const float (&e)[K], const bool (&skip)[K])
{
    static_assert(K % 2 == 0, "s = lane*K + k: k even <=> blank state");
    float pre[K];
    // blank states first (k even): two predecessors, no skip.  States beyond n carry the sentinel emission and just
    // sink (stay finite: they lose 1e30 per step, fp32 holds that for any T); no per-state masking or clamping.
    // (the states are taken in pairs (k, k + 2) through lse2_2x2; the same arithmetic, operation by operation, as
    // lse2_2 -- the workers' recomputed rows go through this same function)
    if (FWD) {
        const float n1 = wave_shr1(a[K - 1], kNegB);          // previous lane's last (label) state
        if constexpr (K == 4 && kPackedStep) {
            const v2f_t pb = lse2_2x2(v2f_t{a[0], a[2]}, v2f_t{n1, a[1]});
            pre[0] = pb.x; pre[2] = pb.y;
            const v2f_t pl = lse2_2x2(v2f_t{a[1], a[3]}, v2f_t{skip[1] ? pre[0] : a[0], skip[3] ? pre[2] : a[2]});
            pre[1] = pl.x; pre[3] = pl.y;
        } else {
#pragma unroll
            for (int k = 0; k < K; k += 2) pre[k] = lse2_2(a[k], k >= 1 ? a[k - 1] : n1);
#pragma unroll
            for (int k = 1; k < K; k += 2) pre[k] = lse2_2(a[k], skip[k] ? pre[k - 1] : a[k - 1]);
        }
    } else {
        const float n1 = wave_shl1(a[0], kNegB);              // next lane's first (blank) state
        if constexpr (K == 4 && kPackedStep) {
            const v2f_t pb = lse2_2x2(v2f_t{a[0], a[2]}, v2f_t{a[1], a[3]});
            pre[0] = pb.x; pre[2] = pb.y;
            const float nb = wave_shl1(pre[0], kNegB);        // ... and what that blank comes from
            const v2f_t pl = lse2_2x2(v2f_t{a[1], a[3]}, v2f_t{skip[1] ? pre[2] : a[2], skip[3] ? nb : n1});
            pre[1] = pl.x; pre[3] = pl.y;
        } else {
#pragma unroll
            for (int k = 0; k < K; k += 2) pre[k] = lse2_2(a[k], a[k + 1]);
            const float nb = wave_shl1(pre[0], kNegB);        // ... and what that blank comes from
#pragma unroll
            for (int k = 1; k < K; k += 2) pre[k] = lse2_2(a[k], skip[k] ? (k + 1 < K ? pre[k + 1] : nb) : (k + 1 < K ? a[k + 1] : n1));
        }
    }
    if constexpr (K == 4 && kPackedStep) {
        const v2f_t s0 = v2f_t{pre[0], pre[1]} + v2f_t{e[0], e[1]}, s1 = v2f_t{pre[2], pre[3]} + v2f_t{e[2], e[3]};
        a[0] = s0.x; a[1] = s0.y; a[2] = s1.x; a[3] = s1.y;
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = pre[k] + e[k];
    }
}

// the first row: the two entry states only
template <int K, bool FWD>
__device__ __forceinline__ void blank_first(float (&a)[K], const float (&e0)[K], int n)
{
    const int s0 = lane_id() * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        const bool entry = FWD ? (s == 0 || s == 1) : (s == n - 1 || s == n - 2);
        a[k] = (entry && s < n) ? e0[k] : kNegB;
    }
}

// Three-launch schedule: one chain over a whole sample, emissions from the workspace (prefetched
// kRingRegs/K rows ahead), rows with their own emission included into the workspace.
template <int K, bool FWD>
__device__ __forceinline__ void blank_chain(const BlankParams &p, int b, int Tb, int L, float (&a)[K])
{
    constexpr int D = kRingRegs / K;                         // emission rows in flight
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1;
    const float *em = p.em + (int64_t)b * p.T * p.NSP + s0;
    float *out = (FWD ? p.al : p.be) + (int64_t)b * p.T * p.NSP + s0;
    bool skip[K], valid[K];
    blank_state_flags<K, FWD>(p, b, n, skip, valid);
    auto row_of = [&](int i) { return FWD ? i : Tb - 1 - i; };
    auto fetch = [&](float (&dst)[K], int i) {
        const float *r = em + (int64_t)row_of(i < Tb ? i : Tb - 1) * p.NSP;
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = r[k];
    };
    auto store = [&](int i) {
        float *r = out + (int64_t)row_of(i) * p.NSP;
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = a[k];
    };
    float ring[D][K];
    {
        float e0[K];
        fetch(e0, 0);
        blank_first<K, FWD>(a, e0, n);
        store(0);
    }
    int i = 1;
#pragma unroll
    for (int j = 0; j < D; ++j) fetch(ring[j], i + j);
    for (; i + D <= Tb; i += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float e[K];
#pragma unroll
            for (int k = 0; k < K; ++k) e[k] = ring[j][k];
            fetch(ring[j], i + j + D);
            blank_step<K, FWD>(a, e, skip);
            store(i + j);
        }
    }
#pragma unroll
    for (int j = 0; j < D; ++j)
        if (i + j < Tb) {
            blank_step<K, FWD>(a, ring[j], skip);
            store(i + j);
        }
}

// ---- fused schedule, the sample's workgroup -------------------------------------------------------
// LDS of a chain workgroup: flags[4 d + j] rows finished by loader j of direction d (0 alpha,
// 1 beta), flags[4 d + 3] steps consumed by that chain; then the two rings of kRingRows/K emission
// rows of NSP floats, row i of a direction in slot i mod ring; then a row of staging per loader.
struct FusedLds {
    int *flags;
    float *ring0;                                            // direction d: ring0 + d * ring_floats
    int ring_floats;
    __device__ __forceinline__ float *ring(int dir) const { return ring0 + dir * ring_floats; }
};
template <int K>
__device__ __forceinline__ FusedLds fused_lds(const BlankParams &p, void *base)
{
    FusedLds f;
    f.flags = reinterpret_cast<int *>(base);
    f.ring0 = reinterpret_cast<float *>(reinterpret_cast<char *>(base) + kFusedLdsHead);
    f.ring_floats = (kRingRows / K) * p.NSP;
    return f;
}

// K consecutive floats of an LDS row, 4 K bytes aligned: one ds_read / ds_write of that width
template <int K>
__device__ __forceinline__ void lds_get(const float *src, float (&dst)[K])
{
    if constexpr (K == 2) {
        const float2 v = *reinterpret_cast<const float2 *>(src);
        dst[0] = v.x; dst[1] = v.y;
    } else {
#pragma unroll
        for (int q = 0; q < K; q += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + q);
            dst[q] = v.x; dst[q + 1] = v.y; dst[q + 2] = v.z; dst[q + 3] = v.w;
        }
    }
}
template <int K>
__device__ __forceinline__ void lds_put(float *dst, const float (&src)[K])
{
    if constexpr (K == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(src[0], src[1]);
    } else {
#pragma unroll
        for (int q = 0; q < K; q += 4) *reinterpret_cast<float4 *>(dst + q) = make_float4(src[q], src[q + 1], src[q + 2], src[q + 3]);
    }
}

// Loader `j` of direction `dir`: rows j, j+kLoaders, ... of that chain's step order, gathered from
// log_probs with kLoadAhead rows in flight, converted to log2 units and written to the ring as
// soon as the chain has left the slot; one counter bump per row.
template <int K>
__device__ __forceinline__ void blank_loader(const BlankParams &p, int b, int Tb, int L, int dir, int j, const FusedLds &f)
{
    constexpr int R = kRingRows / K, P = kLoadAhead;
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1;
    const float *base = p.lp + (int64_t)b * p.sb;
    int c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) c[k] = p.cls[b * p.NSP + s0 + k];
    float *ring = f.ring(dir) + s0;
    int *done_flag = f.flags + 4 * dir + j;
    const int *consumed = f.flags + 4 * dir + 3;
    int seen = 0;                                            // steps the chain is known to have consumed
    auto issue = [&](float (&x)[K], int r) {                 // row r of the step order (clamped: loaded, not used)
        const int rr = r < Tb ? r : Tb - 1;
        const float *row = base + (int64_t)(dir == 0 ? rr : Tb - 1 - rr) * p.st;
#pragma unroll
        for (int k = 0; k < K; ++k) x[k] = row[c[k]];
    };
    float x[P][K];
#pragma unroll
    for (int q = 0; q < P; ++q) issue(x[q], j + kLoaders * q);
    int done = 0;
    for (int r = j; r < Tb; r += kLoaders * P) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int rr = r + kLoaders * q;
            if (rr < Tb) {                                   // wave-uniform
                if (rr - R + 1 > seen && !lds_wait_ge(p, consumed, rr - R + 1, seen)) return;   // slot still being read
                float e[K];
#pragma unroll
                for (int k = 0; k < K; ++k) e[k] = s0 + k < n ? fmaxf(x[q][k] * kLog2e, kNegB) : kNegB;
                lds_put<K>(ring + (rr & (R - 1)) * p.NSP, e);
                ++done;                                      // (LDS keeps a wave's program order: row, then count)
                if (lane == 0) wg_store(done_flag, done);
            }
            issue(x[q], rr + kLoaders * P);
        }
    }
}

// The same with whole rows: a gather spreads a wave's 64 lanes over up to 64 cache lines and the
// CU's address path takes them one by one -- two directions at one row per step keep it busy
// ~95 % of the time and the chains wait.  Here the row comes in as C/4 coalesced float4 (VEC4
// shapes: C <= 1024, 16-byte aligned rows), goes through this wave's `stage` buffer in LDS and
// is gathered from there.
template <int K>
__device__ __forceinline__ void blank_loader_rows(const BlankParams &p, int b, int Tb, int L, int dir, int j, const FusedLds &f, float *stage)
{
    constexpr int R = kRingRows / K, P = kLoadAhead;
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1, c4 = p.C >> 2;
    const float *base = p.lp + (int64_t)b * p.sb;
    int c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) c[k] = p.cls[b * p.NSP + s0 + k];
    float *ring = f.ring(dir) + s0;
    int *done_flag = f.flags + 4 * dir + j;
    const int *consumed = f.flags + 4 * dir + 3;
    int seen = 0;                                            // steps the chain is known to have consumed
    // Steps [0, H) of this direction are gathered here (its chain leaves them in the workspace as well); steps >= H are rows
    // the OTHER direction gathered in its first half: they come back from the workspace, K floats per lane.  Two loops, each
    // with a FIXED number of memory operations per row in flight: with one loop and a branch per row the compiler must
    // assume the fewest operations behind a row it waits for, and the twelve rows in flight shrink to three.
    const int H = kGatherOnce ? (Tb + 1) >> 1 : Tb;
    const __amdgpu_buffer_rsrc_t ersrc = lattice_rsrc(p.em + (int64_t)b * p.T * p.NSP, p.T, p.NSP);
    const int lane_off = s0 < n ? s0 * (int)sizeof(float) : kPastLattice;
    auto issue = [&](f4_t (&x)[kMaxV4], int r) {             // row r of the step order (clamped: loaded, not used)
        const int rr = r < H ? r : H - 1;
        const f4_t *row = reinterpret_cast<const f4_t *>(base + (int64_t)(dir == 0 ? rr : Tb - 1 - rr) * p.st);
#pragma unroll
        for (int q = 0; q < kMaxV4; ++q) x[q] = row[min(lane + kWave * q, c4 - 1)];   // (past the row: its last float4 again)
    };
    auto put = [&](const float (&e)[K], int rr, int &done) { // false: the wait ran out
        if (rr - R + 1 > seen && !lds_wait_ge(p, consumed, rr - R + 1, seen)) return false;   // slot still being read
        lds_put<K>(ring + (rr & (R - 1)) * p.NSP, e);
        ++done;                                              // (LDS keeps a wave's program order: row, then count)
        if (lane == 0) wg_store(done_flag, done);
        return true;
    };
    auto finish = [&](const f4_t (&x)[kMaxV4], int rr, int &done) {
#pragma unroll
        for (int v = 0; v < kMaxV4; ++v) reinterpret_cast<f4_t *>(stage)[lane + kWave * v] = x[v];   // (stage holds 4 x 64 float4)
        asm volatile("" ::: "memory");                       // (same wave: LDS keeps program order)
        float e[K];
#pragma unroll
        for (int k = 0; k < K; ++k) e[k] = s0 + k < n ? fmaxf(stage[c[k]] * kLog2e, kNegB) : kNegB;
        return put(e, rr, done);
    };
    // kLoadAhead rows in flight, each in its own small array: one big array would stay in scratch
    // memory (the backend only keeps arrays up to a quarter of the register budget in registers)
    static_assert(P == 12, "one named buffer per row in flight");
    int done = 0;
    if (j < H) {
        f4_t x0[kMaxV4], x1[kMaxV4], x2[kMaxV4], x3[kMaxV4], x4[kMaxV4], x5[kMaxV4], x6[kMaxV4], x7[kMaxV4], x8[kMaxV4],
            x9[kMaxV4], x10[kMaxV4], x11[kMaxV4];
#define CTC_EACH_ROW(F) F(0, x0) F(1, x1) F(2, x2) F(3, x3) F(4, x4) F(5, x5) F(6, x6) F(7, x7) F(8, x8) F(9, x9) F(10, x10) F(11, x11)
#define CTC_FIRST(Q, X) issue(X, j + kLoaders * (Q));
#define CTC_TURN(Q, X)                                        \
    {                                                         \
        const int rr = r + kLoaders * (Q);                    \
        if (rr < H && !finish(X, rr, done)) return;           \
        issue(X, rr + kLoaders * P);                          \
    }
        CTC_EACH_ROW(CTC_FIRST)
        for (int r = j; r < H; r += kLoaders * P) { CTC_EACH_ROW(CTC_TURN) }
#undef CTC_TURN
#undef CTC_FIRST
#undef CTC_EACH_ROW
    }
    if (!kGatherOnce) return;
    const int r0 = H + (j + kLoaders - H % kLoaders) % kLoaders;   // this loader's first row >= H (rows r = j mod kLoaders)
    if (r0 >= Tb) return;
    {
        // Row r of our step order is the other direction's step Tb-1-r; its CHAIN stored the emission row and publishes
        // how many of its steps have landed.  Later rows need fewer of them: one wait, for our first row.
        int seen_landed = 0;
        if (!lds_wait_ge(p, f.flags + 10 + (1 - dir), Tb - r0, seen_landed)) return;
    }
    auto fetch = [&](float (&y)[K], int r) {                 // sc0: past the L1 -- the CU's L2 has the row (same workgroup wrote it)
        const int rr = r < Tb ? r : Tb - 1;
        agent_load_row<K, 1>(ersrc, (dir == 0 ? rr : Tb - 1 - rr) * p.NSP * (int)sizeof(float) + lane_off, y);
    };
    float y[P][K];
#pragma unroll
    for (int q = 0; q < P; ++q) fetch(y[q], r0 + kLoaders * q);
    for (int r = r0; r < Tb; r += kLoaders * P) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int rr = r + kLoaders * q;
            if (rr < Tb) {                                   // wave-uniform
                float e[K];
#pragma unroll
                for (int k = 0; k < K; ++k) e[k] = s0 + k < n ? y[q][k] : kNegB;   // (lanes past the states read 0)
                if (!put(e, rr, done)) return;
            }
            fetch(y[q], rr + kLoaders * P);
        }
    }
}

// One chain of the fused schedule: emissions from the LDS ring, kGroup steps per hand-off check;
// lattice rows written through to memory (beta WITHOUT its own emission), and the number of steps
// whose rows have landed published per group.
// ---- pool gather (see kPoolGather) -------------------------------------------------------------------------------
// rows of chunk c of a sample with T_b frames the pool will gather: distances [c kPoolChunk, min(.., H)), both ends,
// the middle row of an odd T_b once
__device__ __forceinline__ int blank_chunk_rows(int Tb, int c)
{
    const int H = (Tb + 1) >> 1, lo = c * kPoolChunk, hi = min(lo + kPoolChunk, H);
    if (hi <= lo) return 0;
    return 2 * (hi - lo) - (((Tb & 1) && H - 1 >= lo && H - 1 < hi) ? 1 : 0);
}

// A worker wave's share of the gather: items q = (distance i, sample b, end) in that order of priority, kPoolBatch
// consecutive items per turn, turns dealt round-robin over the nw worker waves.  Never waits for anybody.
template <int K>
__device__ __forceinline__ void blank_pool_gather(const BlankParams &p, int wid, int nw, float *stage)
{
    constexpr int NB = kPoolBatch;
    const int lane = lane_id(), s0 = lane * K, c4 = p.C >> 2;
    const long long Q = 2ll * p.B * ((p.T + 1) >> 1);
    for (long long q0 = (long long)wid * NB; q0 < Q; q0 += (long long)nw * NB) {
        int bq[NB], tq[NB], nq[NB], cq[NB];
        f4_t x[NB][kMaxV4];
        i4_t cl[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {                       // (wave-uniform bookkeeping; every item issues the same loads)
            const long long q = q0 + k;
            const int i = (int)(q / (2 * p.B)), r = (int)(q - (long long)i * 2 * p.B);
            const int b = r >> 1, end = r & 1;
            int Tb, Lb;                                      // (the lengths the sample's own workgroup goes by: a sample
            blank_sample_ok(p, b, Tb, Lb);                   // without an alignment still runs its chains, to -inf)
            const int H = (Tb + 1) >> 1;
            const bool live = q < Q && i < H && !(end == 1 && (Tb & 1) && i == H - 1);
            bq[k] = __builtin_amdgcn_readfirstlane(b);
            tq[k] = __builtin_amdgcn_readfirstlane(live ? (end ? Tb - 1 - i : i) : -1);
            nq[k] = __builtin_amdgcn_readfirstlane(2 * Lb + 1);
            cq[k] = __builtin_amdgcn_readfirstlane(i / kPoolChunk);
            const f4_t *row = reinterpret_cast<const f4_t *>(p.lp + (int64_t)(tq[k] >= 0 ? tq[k] : 0) * p.st + (int64_t)bq[k] * p.sb);
#pragma unroll
            for (int v = 0; v < kMaxV4; ++v) x[k][v] = row[min(lane + kWave * v, c4 - 1)];   // (past the row: its last float4 again)
            static_assert(K == 4 || K == 2 || K == 8, "class table read below");
            if constexpr (K == 4) cl[k] = *reinterpret_cast<const i4_t *>(p.cls + bq[k] * p.NSP + s0);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (tq[k] < 0) continue;                         // (wave-uniform)
#pragma unroll
            for (int v = 0; v < kMaxV4; ++v) reinterpret_cast<f4_t *>(stage)[lane + kWave * v] = x[k][v];
            asm volatile("" ::: "memory");                   // (same wave: LDS keeps program order)
            float e[K];
            if constexpr (K == 4) {
                const int c0 = cl[k].x, c1 = cl[k].y, c2 = cl[k].z, c3 = cl[k].w;
                const int cc[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int j = 0; j < K; ++j) e[j] = s0 + j < nq[k] ? fmaxf(stage[cc[j]] * kLog2e, kNegB) : kNegB;
            } else {
#pragma unroll
                for (int j = 0; j < K; ++j) e[j] = s0 + j < nq[k] ? fmaxf(stage[p.cls[bq[k] * p.NSP + s0 + j]] * kLog2e, kNegB) : kNegB;
            }
            asm volatile("" ::: "memory");                   // (the staged row is read before the next one overwrites it)
            const __amdgpu_buffer_rsrc_t ersrc = lattice_rsrc(p.em + (int64_t)bq[k] * p.T * p.NSP, p.T, p.NSP);
            agent_store_row<K>(ersrc, tq[k] * p.NSP * (int)sizeof(float) + (s0 < nq[k] ? s0 * (int)sizeof(float) : kPastLattice), e);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the rows have landed ...
        if (lane == 0) {                                     // ... before they are counted
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (tq[k] >= 0)
                    __hip_atomic_fetch_add(p.chunk + bq[k] * p.nchunk + cq[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the scout of a direction: polls the sample's chunk counters in order, publishes "chunks complete" in LDS
__device__ __forceinline__ void blank_pool_scout(const BlankParams &p, int b, int Tb, int *verified)
{
    const int H = (Tb + 1) >> 1, nch = (H + kPoolChunk - 1) / kPoolChunk;
    const int *cnt = p.chunk + b * p.nchunk;
    for (int c = 0; c < nch; ++c) {
        const int want = blank_chunk_rows(Tb, c);
        bool ok = false;
        for (int it = 0; it < kSpinLimit; ++it) {
            if (agent_load(cnt + c) >= want) { ok = true; break; }
            if ((it & 255) == 255 && agent_load(p.sync) != 0) break;   // somebody already gave up
            __builtin_amdgcn_s_sleep(20);
        }
        if (!ok) { agent_store(p.sync, 1); raise_status(p.counter, kStatusBlankStarved); return; }   // (the loaders' waits run out)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane_id() == 0) wg_store(verified, c + 1);
    }
}

// data loader j (of kPoolLoaders) of direction dir: rows j, j + kPoolLoaders, ... of the chain's step order, K floats per
// lane from the workspace where the pool left them, kLoadAhead rows in flight, into the ring as soon as the chain has left
// the slot.  A row's load is ISSUED only once its chunk is known complete.
template <int K>
__device__ __forceinline__ void blank_pool_loader(const BlankParams &p, int b, int Tb, int L, int dir, int j, const FusedLds &f)
{
    constexpr int R = kRingRows / K, P = kLoadAhead, NL = kPoolLoaders;
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1;
    float *ring = f.ring(dir) + s0;
    int *done_flag = f.flags + 4 * dir + j;
    const int *consumed = f.flags + 4 * dir + 3;
    const int *verified = f.flags + 12 + dir;
    const __amdgpu_buffer_rsrc_t ersrc = lattice_rsrc(p.em + (int64_t)b * p.T * p.NSP, p.T, p.NSP);
    const int lane_off = s0 < n ? s0 * (int)sizeof(float) : kPastLattice;
    int seen = 0, chunks = 0, done = 0;
    bool dead = false;                                       // a bounded wait ran out (never observed): stop, say so
    auto fetch = [&](float (&y)[K], int r) {                 // row r of the step order (past the end: the last row again)
        const int rr = r < Tb ? r : Tb - 1;
        const int c = min(rr, Tb - 1 - rr) / kPoolChunk;     // the chunk the pool counts this row in
        if (c >= chunks && !dead && !lds_wait_ge_quiet(verified, c + 1, chunks)) dead = true;
        agent_load_row<K>(ersrc, (dir == 0 ? rr : Tb - 1 - rr) * p.NSP * (int)sizeof(float) + lane_off, y);
    };
    float y[P][K];
#pragma unroll
    for (int q = 0; q < P; ++q) fetch(y[q], j + NL * q);
    for (int r = j; r < Tb && !dead; r += NL * P) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int rr = r + NL * q;
            if (rr < Tb && !dead) {                          // wave-uniform
                float e[K];
#pragma unroll
                for (int k = 0; k < K; ++k) e[k] = s0 + k < n ? y[q][k] : kNegB;   // (lanes past the states read 0)
                if (rr - R + 1 > seen && !lds_wait_ge_quiet(consumed, rr - R + 1, seen)) dead = true;   // slot still being read
                lds_put<K>(ring + (rr & (R - 1)) * p.NSP, e);
                ++done;                                      // (LDS keeps a wave's program order: row, then count)
                if (lane == 0 && !dead) wg_store(done_flag, done);
            }
            fetch(y[q], rr + NL * P);
        }
    }
    if (dead) { agent_store(p.sync, 1); raise_status(p.counter, kStatusBlankStarved); }
}

template <int K, bool FWD, int NL>
__device__ __forceinline__ void blank_chain_fused(const BlankParams &p, int b, int Tb, int L, float (&a)[K], const FusedLds &f, bool once)
{
    constexpr int kLoaders = NL;                             // data loader waves of this direction (shadows the global)
    constexpr int R = kRingRows / K, G = kGroup < R / 2 ? kGroup : R / 2;
    static_assert(G <= R / 2, "a group must fit in the ring twice");
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1, dir = FWD ? 0 : 1;
    const __amdgpu_buffer_rsrc_t orsrc = lattice_rsrc((FWD ? p.al : p.be) + (int64_t)b * p.T * p.NSP, p.T, p.NSP);
    int *prog = p.sync + kSyncHead + (dir * p.Bp + b) * kProgPitch;
    const float *ring = f.ring(dir) + s0;
    const int *loaded = f.flags + 4 * dir;
    int *consumed = f.flags + 4 * dir + 3;
    bool skip[K], valid[K], starved = false;
    blank_state_flags<K, FWD>(p, b, n, skip, valid);
    // rows <= last of the step order are in the ring: loader j has then finished (last - j)/kLoaders + 1 rows
    int have[kLoaders] = {};
    unsigned long long waited = 0, polls = 0;                // (diagnostics)
#ifdef CTC_AMD_DIAGNOSTICS
    const bool probe = FWD && b == 0 && (p.debug & 256);     // (diagnostics build: per-group cycle breakdown, see the main loop)
#else
    constexpr bool probe = false;
#endif
    unsigned long long pr[2][6] = {};
    auto need_rows = [&](int last) {
        int need[kLoaders];
        bool all = true;
#pragma unroll
        for (int j = 0; j < kLoaders; ++j) {
            need[j] = last >= j ? (last - j) / kLoaders + 1 : 0;
            all = all && need[j] <= have[j];
        }
        if (all) return;
#pragma unroll
        for (int j = 0; j < kLoaders; ++j) have[j] = wg_load(loaded + j);   // (one LDS round trip for the three flags)
#pragma unroll
        for (int j = 0; j < kLoaders; ++j) {
            if (need[j] > have[j]) {
                const unsigned long long t0 = (p.debug & 128) ? __builtin_amdgcn_s_memtime() : 0;
                if (!lds_wait_ge(p, loaded + j, need[j], have[j])) starved = true;
                if (p.debug & 128) { waited += __builtin_amdgcn_s_memtime() - t0; ++polls; }
            }
        }
    };
    auto em_row = [&](float (&dst)[K], int i) {
        lds_get<K>(ring + (i & (R - 1)) * p.NSP, dst);
    };
    // lanes whose K states all lie beyond the sample's n states get an offset past the end of the buffer:
    // the hardware drops their part of the store (config 5: 13 of 64 lanes, a fifth of the lattice bytes)
    const int lane_off = s0 < n ? s0 * (int)sizeof(float) : kPastLattice;
    // gather-once: the emission rows of the first-half steps go to the workspace for the other direction's second half
    const int H = once ? (Tb + 1) >> 1 : 0;
    const __amdgpu_buffer_rsrc_t ersrc = lattice_rsrc(p.em + (int64_t)b * p.T * p.NSP, p.T, p.NSP);
    int *landed = f.flags + 10 + dir;
    auto store = [&](int i, const float (&e)[K]) {
        const int t = FWD ? i : Tb - 1 - i;
        if (i < H) agent_store_row<K, 0>(ersrc, t * p.NSP * (int)sizeof(float) + lane_off, e);   // (wave-uniform)
        if (kHalfLattice && (t & 1) != (FWD ? 0 : 1)) return;   // (wave-uniform) the workers recompute this row
        const int off = t * p.NSP * (int)sizeof(float) + lane_off;
        if (FWD) {
            agent_store_row<K>(orsrc, off, a);
        } else {
            float v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = a[k] - e[k];
            agent_store_row<K>(orsrc, off, v);
        }
    };
    {
        need_rows(0);
        float e0[K];
        em_row(e0, 0);
        blank_first<K, FWD>(a, e0, n);
        store(0, e0);
    }
    // Two segments when the emission rows are shared (gather-once): steps [1, H) need only this direction's own
    // loaders; before the first step >= H everything stored so far is drained and published -- the other direction's
    // second half waits for exactly these rows, and with the usual lag both chains would wait for each other at H.
    int i = 1;
#pragma nounroll
    for (int seg = 0; seg < 2; ++seg) {
        const int hi = (seg == 0 && H > 0) ? (H < Tb ? H : Tb) : Tb;
        for (; i + G <= hi; i += G) {
            if (FWD && b == 0 && (i - 1) % 256 < G) bstamp(p, 1 + (i - 1) / 256);
            // (diagnostics, CTC_AMD_BLANK_DEBUG & 256: where a group of G steps of sample 0's alpha chain spends its cycles)
            const unsigned long long q0 = probe ? __builtin_amdgcn_s_memtime() : 0;
            need_rows(i + G - 1);
            const unsigned long long q1 = probe ? __builtin_amdgcn_s_memtime() : 0;
            float e[G][K];
#pragma unroll
            for (int j = 0; j < G; ++j) em_row(e[j], i + j);
            if (probe) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long q2 = probe ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
            for (int j = 0; j < G; ++j) {
                blank_step<K, FWD>(a, e[j], skip);
                store(i + j, e[j]);
            }
            const unsigned long long q3 = probe ? __builtin_amdgcn_s_memtime() : 0;
            if (lane == 0) wg_store(consumed, i + G);        // the loaders may refill these slots
            // Only stores go through this wave's vector-memory counter, it retires in order, and a step
            // issues at least one: at most kLandLag outstanding => the rows of the steps before
            // i + G - kLandLag have landed.
            // (every other step stores when only half of the lattice is kept: half as many may be outstanding; groups that
            // also store their emission rows have kLandLag more in the last kLandLag steps)
            if (i + G <= H) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kHalfLattice ? kLandLag / 2 : kLandLag) + kLandLag) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kHalfLattice ? kLandLag / 2 : kLandLag) : "memory");
            const unsigned long long q4 = probe ? __builtin_amdgcn_s_memtime() : 0;
            if (lane == 0 && i + G > kLandLag) {
                agent_store(prog, i + G - kLandLag);
                if (once) wg_store(landed, i + G - kLandLag);
            }
            if (probe) {
                const unsigned long long q5 = __builtin_amdgcn_s_memtime();
                const int half = 2 * i >= Tb ? 1 : 0;         // first half: worker pool idle; second: beside the streaming workers
                pr[half][0] += q1 - q0; pr[half][1] += q2 - q1; pr[half][2] += q3 - q2; pr[half][3] += q4 - q3; pr[half][4] += q5 - q4;
                pr[half][5] += 1;
            }
        }
        if (i < hi) {
            need_rows(hi - 1);
            for (; i < hi; ++i) {
                float e[K];
                em_row(e, i);
                blank_step<K, FWD>(a, e, skip);
                store(i, e);
            }
            if (lane == 0 && hi < Tb) wg_store(consumed, i);
        }
        if (seg == 0 && H > 0 && hi < Tb) {                  // the end of the first half: drain, publish
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                agent_store(prog, hi);
                wg_store(landed, hi);
            }
        }
    }
    if (lane == 0) wg_store(consumed, Tb + R);               // (nothing left to protect)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (FWD && b == 0) bstamp(p, 10);
    if (FWD && b == 0 && (p.debug & 128) && lane == 0) {
        reinterpret_cast<unsigned long long *>(p.counter)[8 + 16] = waited;
        reinterpret_cast<unsigned long long *>(p.counter)[8 + 17] = polls;
    }
    if (probe && lane == 0) {                                // (the probe reuses the timeline's slots 0..11: run them apart)
        for (int h = 0; h < 2; ++h)
            for (int k = 0; k < 6; ++k) reinterpret_cast<unsigned long long *>(p.counter)[8 + 6 * h + k] = pr[h][k];
    }
    if (lane == 0) agent_store(prog, starved ? -1 : Tb);     // (a starved chain never releases its rows)
    if (lane == 0 && once && !starved) wg_store(landed, Tb);
    if (starved) a[0] = __builtin_nanf("");
}

// likelihood of sample b from the last alpha row, and the batch mean
template <int K>
__device__ __forceinline__ void blank_publish(const BlankParams &p, int b, bool ok, int Tb, int L, const float (&a)[K])
{
    const int n = 2 * L + 1;
    float nll = __builtin_inff();
    if (ok && Tb > 0) {
        float v1 = 0.f, v2 = 0.f;                            // alpha_{T-1}(n-1), alpha_{T-1}(n-2)
        bool bad = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int s = lane_id() * K + k;
            if (s == n - 1) v1 = a[k];
            if (s == n - 2) v2 = a[k];
            bad = bad || a[k] != a[k];
        }
        v1 = wave_sum(v1);
        v2 = n >= 2 ? wave_sum(v2) : kNegB;
        const float ll2 = lse2_2(v1, v2);
        nll = ll2 < -1.0e29f ? __builtin_inff() : -ll2 * kLn2;
        if (__builtin_amdgcn_ballot_w64(bad) != 0) nll = __builtin_nanf("");   // (fused: a wait ran out)
    } else if (ok && L == 0) {
        nll = 0.f;                                           // empty input, empty target
    }
    publish_and_reduce(nll, b, p.B, p.nll, p.loss, p.loss_scale, p.counter,
                       [&](float v, int i) {
                           const int64_t Li = p.tgt_len[i];
                           return v / (float)(Li > 1 ? Li : 1);
                       });
}

// three-launch schedule: wave 0 = alpha (+ likelihood and batch mean), wave 1 = beta
template <int K>
__global__ __launch_bounds__(128) void blank_chain_kernel(BlankParams p)
{
    const int b = blockIdx.x, w = wave_id();
    int Tb, L;
    const bool ok = blank_sample_ok(p, b, Tb, L);
    float a[K];
    if (w == 0) {
        if (ok && Tb > 0) blank_chain<K, true>(p, b, Tb, L, a);
        blank_publish<K>(p, b, ok, Tb, L, a);
    } else if (w == 1 && p.grad && ok && Tb > 0) {
        blank_chain<K, false>(p, b, Tb, L, a);
    }
}

// ---- K2: gamma -> gradient rows ----------------------------------------------------------------
constexpr int kGradWaves = 4;


// One (t,b) row of work for a wave: everything it loads from HBM.
template <int K>
struct BlankRow {
    float4 xr[kMaxV4];
    float al[K], be[K], em[K];
    int t, b, Tb, L;
    bool live;                                               // false: a zero row (beyond T_b, or no alignment)
    bool poison;                                             // fused schedule: the wait for the chains ran out
};

// SYNC = false (three launches): idx = t B + b, the lattice is complete.
// SYNC = true (fused): idx = (2 m + side) B + b names the row at distance m from the FAR end of
// its sample -- side 0 is t = m (a real row while m >= T_b-1-m, a zero row from T_b on), side 1
// is t = T_b-1-m (while 0 <= t < m) -- i.e. the rows in the order in which they get both alpha
// and beta; the loads wait until both chains have published the row; beta comes without its
// emission, so none is loaded.  r.t < 0: no row here.
// How far the two chains of a sample were when this wave last looked.  A look is a round trip to memory
// (two when done one after the other) in front of a row's own loads, and the poll retires behind every
// load the wave has in flight: with a look per row the worker pool did 310 rows/us.  A wave mostly stays
// with one sample and the chains run ahead of a busy worker, so most rows need no look at all.
struct ChainsSeen {
    int b = -1, fwd = 0, bwd = 0;
};

// both chains of sample b have landed `need_f` / `need_b` steps; false when the bounded wait ran out
__device__ __forceinline__ bool wait_chains(const BlankParams &p, int b, int need_f, int need_b, ChainsSeen &ps)
{
    if (ps.b != b) { ps.b = b; ps.fwd = ps.bwd = 0; }        // wave-uniform
    if (ps.fwd >= need_f && ps.bwd >= need_b) return true;
    const int *pf = p.sync + kSyncHead + b * kProgPitch, *pb = pf + p.Bp * kProgPitch;
    bool ok = false;
    for (int it = 0; it < kSpinLimit; ++it) {
        const int vf = agent_load(pf), vb = agent_load(pb);  // (one round trip for the two)
        ps.fwd = vf; ps.bwd = vb;
        if (vf >= need_f && vb >= need_b) { ok = true; break; }
        if (vf < 0 || vb < 0 || ((it & 255) == 255 && agent_load(p.sync) != 0)) break;   // a chain gave up / somebody did
        const int naps = max(min(max(need_f - vf, need_b - vb), 64), 24);   // ~ a nap (1 us) per 8 missing steps, 3 to 8
        for (int q = 0; q < naps; q += 8) __builtin_amdgcn_s_sleep(40);
    }
    if (!ok) { agent_store(p.sync, 1); raise_status(p.counter, kStatusBlankStarved); }
    asm volatile("" ::: "memory");                           // nothing below moves above the poll
    return ok;
}

template <int K, bool VEC4, bool SYNC>
__device__ __forceinline__ void blank_row_load(const BlankParams &p, int idx, BlankRow<K> &r, ChainsSeen &ps)
{
    const int lane = lane_id();
    const int q = idx / p.B;
    r.b = __builtin_amdgcn_readfirstlane(idx - q * p.B);     // consecutive waves -> consecutive b: contiguous rows
    // (through the scalar cache, and the SAME loads on every path -- dead rows and indices past the end load a clamped
    // row and ignore it: see blank_pair_load on what a vector load of the lengths or an early exit costs the rows in flight)
    int2 meta;
    {
        unsigned long long bits;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bits) : "s"(p.meta + r.b) : "memory");
        meta.x = (int)(unsigned)bits;
        meta.y = (int)(bits >> 32);
    }
    const int Te = meta.x;                                   // 0 without an alignment: every row is a zero row
    r.Tb = Te;
    r.L = meta.y;
    r.poison = false;
    if (!SYNC) {
        r.t = q < p.T ? q : -1;
    } else {
        const int m = q >> 1;
        if ((q & 1) == 0) r.t = (m < p.T && (m >= Te || 2 * m >= Te - 1)) ? m : -1;
        else r.t = (m < Te && Te - 1 - m < m) ? Te - 1 - m : -1;
        r.t = __builtin_amdgcn_readfirstlane(r.t);
    }
    r.live = r.t >= 0 && r.t < Te;
    const int te = r.live ? r.t : 0;
    // (lanes beyond the sample's states: past the end of the buffer, they load nothing and read 0)
    const int off = te * p.NSP * (int)sizeof(float) + (lane * K < 2 * r.L + 1 ? lane * K * (int)sizeof(float) : kPastLattice);
    if (SYNC) {
        if (r.live && !wait_chains(p, r.b, r.t + 1, Te - r.t, ps)) r.poison = true;
        agent_load_row<K>(lattice_rsrc(p.al + (int64_t)r.b * p.T * p.NSP, p.T, p.NSP), off, r.al);
        agent_load_row<K>(lattice_rsrc(p.be + (int64_t)r.b * p.T * p.NSP, p.T, p.NSP), off, r.be);
    }
    const int64_t o = ((int64_t)r.b * p.T + te) * p.NSP + lane * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!SYNC) { r.al[k] = p.al[o + k]; r.be[k] = p.be[o + k]; r.em[k] = p.em[o + k]; }
    }
    if (VEC4) {
        const float4 *row = reinterpret_cast<const float4 *>(p.lp + (int64_t)te * p.st + (int64_t)r.b * p.sb);
        const int c4 = p.C >> 2;
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) r.xr[i] = row[min(lane + kWave * i, c4 - 1)];   // (past the row: its last float4 again)
    }
}

// The state tables of the sample a wave is working on: this lane's K states in registers, the
// repeat chain `nxt` in the wave's LDS (walked per row: from global memory every hop was a
// dependent L2 round trip on the row's critical path).  Reloaded when the sample changes; a
// wave's stride over the rows is usually a multiple of B, then that is once.
template <int K>
struct BlankTables {
    int b = -1;
    int cls[K], nxt[K];
    bool first[K];
    bool skipf[K], skipb[K];                                 // (row pairs) the s-2 / s+2 edge of a forward / backward step
};
template <int K>
__device__ __forceinline__ void blank_tables_for(const BlankParams &p, int b, BlankTables<K> &tb, int *nxt_l)
{
    if (tb.b == b) return;                                   // wave-uniform
    tb.b = b;
    const int s0 = lane_id() * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        tb.cls[k] = p.cls[b * p.NSP + s0 + k];
        tb.nxt[k] = p.nxt[b * p.NSP + s0 + k];
        tb.first[k] = p.first[b * p.NSP + s0 + k] != 0;
        nxt_l[s0 + k] = tb.nxt[k];
    }
    if (kHalfLattice) {
        bool valid[K];
        const int n = 2 * p.meta[b].y + 1;
        blank_state_flags<K, true>(p, b, n, tb.skipf, valid);
        blank_state_flags<K, false>(p, b, n, tb.skipb, valid);
    }
}

template <int K, bool VEC4, bool SYNC>
__device__ __forceinline__ void blank_row_finish(const BlankParams &p, const BlankRow<K> &r, float *occ, float *gam, BlankTables<K> &tb)
{
    const int lane = lane_id(), s0 = lane * K;
    int *nxt_l = reinterpret_cast<int *>(gam + p.NSP);
    if (r.t < 0) return;                                     // (fused) this index names no row
    float *g = p.grad + ((int64_t)r.t * p.B + r.b) * p.C;
    if (!r.live || r.poison) {
        const float z = r.poison ? __builtin_nanf("") : 0.f;
        if (VEC4) {
            for (int q = lane; q < (p.C >> 2); q += kWave) stream_store(reinterpret_cast<float4 *>(g) + q, make_float4(z, z, z, z));
        } else {
            for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], z);
        }
        return;
    }
    const int n = 2 * r.L + 1;
    blank_tables_for<K>(p, r.b, tb, nxt_l);
    float v[K];
    float m = kNegB;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? (SYNC ? r.al[k] + r.be[k] : r.al[k] + r.be[k] - r.em[k]) : kNegB;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float ssum = 0.f, blank_part = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? __builtin_amdgcn_exp2f(v[k] - m) : 0.f;     // lattice is in log2 units
        ssum += v[k];
        if (((s0 + k) & 1) == 0) blank_part += v[k];
    }
    ssum = wave_sum(ssum);
    blank_part = wave_sum(blank_part);
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int k = 0; k < K; ++k) gam[s0 + k] = v[k] * inv;            // wave-local LDS, in order
    // occupancy per class: blank from the reduction, labels folded along the repeat chain
    if (lane == 0) occ[p.blank] = blank_part * inv;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (tb.first[k]) {                                    // label states only; repeats are chained
            float tot = gam[s0 + k];
            for (int q = tb.nxt[k]; q >= 0; q = nxt_l[q]) tot += gam[q];
            occ[tb.cls[k]] = tot;
        }
    }
    const float gs = p.grad_scale / (float)(r.L > 1 ? r.L : 1);
    if (VEC4) {
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) {
            const int q = lane + kWave * i;
            if (q < (p.C >> 2)) {
                const float4 o = reinterpret_cast<const float4 *>(occ)[q];
                float4 out;
                out.x = (fast_exp(r.xr[i].x) - o.x) * gs;
                out.y = (fast_exp(r.xr[i].y) - o.y) * gs;
                out.z = (fast_exp(r.xr[i].z) - o.z) * gs;
                out.w = (fast_exp(r.xr[i].w) - o.w) * gs;
                stream_store(reinterpret_cast<float4 *>(g) + q, out);
            }
        }
    } else {
        const float *row = p.lp + (int64_t)r.t * p.st + (int64_t)r.b * p.sb;
        for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], (fast_exp(row[c]) - occ[c]) * gs);
    }
    // un-set only what this row touched
    if (lane == 0) occ[p.blank] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        if ((s & 1) && s < n) occ[tb.cls[k]] = 0.f;
    }
}

// ---- row PAIRS (persistent launch with half of the lattice in memory) -----------------------------------
// A pair is rows t = 2P and t + 1 of a sample.  Memory holds alpha_t and beta'_{t+1}; with e = the emissions of
// row t + 1 gathered from the log_probs row the wave holds anyway:
//     alpha_{t+1} = step_fwd(alpha_t) + e          (bit-identical to the chain's own row)
//     beta'_t     = step_bwd(beta'_{t+1} + e)      (the chain's beta_{t+1} up to one rounding)
// The last row of a sample with an odd number of rows has no partner: beta' there is the entry condition.
template <int K>
struct BlankPair {
    float4 x0[kMaxV4], x1[kMaxV4];
    float al[K], be[K];
    int t, b, Tb, L;                                         // t: the even row, < 0: no pair here
    bool live0, live1, exist1;                               // rows t / t+1 inside T_b; row t+1 inside T
    bool poison;
};

// idx = (2 m + side) B + b over PAIR distances m, the analogue of blank_row_load<SYNC>'s order
template <int K, bool VEC4>
__device__ __forceinline__ void blank_pair_load(const BlankParams &p, int idx, BlankPair<K> &r, ChainsSeen &ps)
{
    const int lane = lane_id();
    const int q = idx / p.B;
    r.b = __builtin_amdgcn_readfirstlane(idx - q * p.B);
    // (through the scalar cache: as a vector load the compiler follows it with s_waitcnt vmcnt(0) -- the value is needed at
    // once -- and that drains every row load this wave has in flight, once per pair)
    int2 meta;
    {
        unsigned long long bits;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bits) : "s"(p.meta + r.b) : "memory");
        meta.x = (int)(unsigned)bits;
        meta.y = (int)(bits >> 32);
    }
    const int Te = meta.x;                                   // 0 without an alignment: every row is a zero row
    r.Tb = Te;
    r.L = meta.y;
    r.poison = false;
    const int Tp = (p.T + 1) >> 1, Tep = (Te + 1) >> 1, m = q >> 1;
    int P;
    if ((q & 1) == 0) P = (m < Tp && (m >= Tep || 2 * m >= Tep - 1)) ? m : -1;
    else P = (m < Tep && Tep - 1 - m < m) ? Tep - 1 - m : -1;
    P = __builtin_amdgcn_readfirstlane(P);
    r.t = P < 0 ? -1 : 2 * P;
    r.live0 = r.t >= 0 && r.t < Te;
    r.live1 = r.t >= 0 && r.t + 1 < Te;
    r.exist1 = r.t >= 0 && r.t + 1 < p.T;
    // The SAME ten loads on every path (dead pairs and missing partners load clamped rows and ignore them): three pairs
    // are in flight per wave, and the compiler turns "wait for pair P" into "at most <loads issued after P's> still
    // outstanding" -- with an early exit or a conditional load on any path that number is zero, every pair waited for
    // ALL loads in flight including the pair just issued, and a wave had one pair's worth of bytes in the air.
    const int te = r.live0 ? r.t : 0, t1 = r.live1 ? r.t + 1 : te;
    const int lane_off = lane * K < 2 * r.L + 1 ? lane * K * (int)sizeof(float) : kPastLattice;
    if (r.live0 && !wait_chains(p, r.b, r.t + 1, r.live1 ? Te - (r.t + 1) : 0, ps)) r.poison = true;
    agent_load_row<K>(lattice_rsrc(p.al + (int64_t)r.b * p.T * p.NSP, p.T, p.NSP), te * p.NSP * (int)sizeof(float) + lane_off, r.al);
    agent_load_row<K>(lattice_rsrc(p.be + (int64_t)r.b * p.T * p.NSP, p.T, p.NSP), t1 * p.NSP * (int)sizeof(float) + lane_off, r.be);
    if (VEC4) {
        const float4 *row0 = reinterpret_cast<const float4 *>(p.lp + (int64_t)te * p.st + (int64_t)r.b * p.sb);
        const float4 *row1 = reinterpret_cast<const float4 *>(p.lp + (int64_t)t1 * p.st + (int64_t)r.b * p.sb);
        const int c4 = p.C >> 2;
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) r.x0[i] = row0[min(lane + kWave * i, c4 - 1)];   // (past the row: its last float4 again)
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) r.x1[i] = row1[min(lane + kWave * i, c4 - 1)];
    }
}

// gamma of one row (v: alpha + beta' in log2 units, this lane's K states) -> its gradient row
template <int K, bool VEC4>
__device__ __forceinline__ void blank_row_emit(const BlankParams &p, int t, int b, int L, float (&v)[K], const float4 (&xr)[kMaxV4],
                                               float *occ, float *gam, const int *nxt_l, const BlankTables<K> &tb)
{
    const int lane = lane_id(), s0 = lane * K, n = 2 * L + 1;
    float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
    float m = kNegB;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? v[k] : kNegB;
        m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float ssum = 0.f, blank_part = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = s0 + k < n ? __builtin_amdgcn_exp2f(v[k] - m) : 0.f;     // lattice is in log2 units
        ssum += v[k];
        if (((s0 + k) & 1) == 0) blank_part += v[k];
    }
    ssum = wave_sum(ssum);
    blank_part = wave_sum(blank_part);
    const float inv = 1.0f / ssum;
#pragma unroll
    for (int k = 0; k < K; ++k) gam[s0 + k] = v[k] * inv;            // wave-local LDS, in order
    if (lane == 0) occ[p.blank] = blank_part * inv;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (tb.first[k]) {                                    // label states only; repeats are chained
            float tot = gam[s0 + k];
            for (int q = tb.nxt[k]; q >= 0; q = nxt_l[q]) tot += gam[q];
            occ[tb.cls[k]] = tot;
        }
    }
    const float gs = p.grad_scale / (float)(L > 1 ? L : 1);
    if (VEC4) {
#pragma unroll
        for (int i = 0; i < kMaxV4; ++i) {
            const int q = lane + kWave * i;
            if (q < (p.C >> 2)) {
                const float4 o = reinterpret_cast<const float4 *>(occ)[q];
                float4 out;
                out.x = (fast_exp(xr[i].x) - o.x) * gs;
                out.y = (fast_exp(xr[i].y) - o.y) * gs;
                out.z = (fast_exp(xr[i].z) - o.z) * gs;
                out.w = (fast_exp(xr[i].w) - o.w) * gs;
                stream_store(reinterpret_cast<float4 *>(g) + q, out);
            }
        }
    } else {
        const float *row = p.lp + (int64_t)t * p.st + (int64_t)b * p.sb;
        for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], (fast_exp(row[c]) - occ[c]) * gs);
    }
    // un-set only what this row touched
    if (lane == 0) occ[p.blank] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int s = s0 + k;
        if ((s & 1) && s < n) occ[tb.cls[k]] = 0.f;
    }
}

template <bool VEC4>
__device__ __forceinline__ void blank_row_fill(const BlankParams &p, int t, int b, float z)
{
    const int lane = lane_id();
    float *g = p.grad + ((int64_t)t * p.B + b) * p.C;
    if (VEC4) {
        for (int q = lane; q < (p.C >> 2); q += kWave) stream_store(reinterpret_cast<float4 *>(g) + q, make_float4(z, z, z, z));
    } else {
        for (int c = lane; c < p.C; c += kWave) stream_store(&g[c], z);
    }
}

template <int K, bool VEC4>
__device__ __forceinline__ void blank_pair_finish(const BlankParams &p, const BlankPair<K> &r, float *occ, float *gam, float *stage,
                                                  BlankTables<K> &tb)
{
    const int lane = lane_id(), s0 = lane * K;
    int *nxt_l = reinterpret_cast<int *>(gam + p.NSP);
    if (r.t < 0) return;                                     // this index names no pair
    if (!r.live0 || r.poison) {
        const float z = r.poison ? __builtin_nanf("") : 0.f;
        blank_row_fill<VEC4>(p, r.t, r.b, z);
        if (r.exist1) blank_row_fill<VEC4>(p, r.t + 1, r.b, z);
        return;
    }
    const int n = 2 * r.L + 1;
    blank_tables_for<K>(p, r.b, tb, nxt_l);
    float a0[K], b0[K];                                      // alpha_t, beta'_t
#pragma unroll
    for (int k = 0; k < K; ++k) a0[k] = s0 + k < n ? r.al[k] : kNegB;
    if (r.live1) {
        float e1[K];                                         // emissions of row t + 1, log2 units (as the loaders make them)
        if (VEC4) {
#pragma unroll
            for (int i = 0; i < kMaxV4; ++i) reinterpret_cast<float4 *>(stage)[lane + kWave * i] = r.x1[i];
            asm volatile("" ::: "memory");                   // (same wave: LDS keeps program order)
#pragma unroll
            for (int k = 0; k < K; ++k) e1[k] = s0 + k < n ? fmaxf(stage[tb.cls[k]] * kLog2e, kNegB) : kNegB;
        } else {
            const float *row1 = p.lp + (int64_t)(r.t + 1) * p.st + (int64_t)r.b * p.sb;
#pragma unroll
            for (int k = 0; k < K; ++k) e1[k] = s0 + k < n ? fmaxf(row1[tb.cls[k]] * kLog2e, kNegB) : kNegB;
        }
        float a1[K], b1[K], zero[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            a1[k] = a0[k];
            b1[k] = s0 + k < n ? r.be[k] : kNegB;            // beta'_{t+1}
            b0[k] = b1[k] + e1[k];                           // beta_{t+1} with its emission
            zero[k] = 0.f;
        }
        blank_step<K, true>(a1, e1, tb.skipf);               // alpha_{t+1}
        blank_step<K, false>(b0, zero, tb.skipb);            // beta'_t
        float v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = a0[k] + b0[k];
        blank_row_emit<K, VEC4>(p, r.t, r.b, r.L, v, r.x0, occ, gam, nxt_l, tb);
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = a1[k] + b1[k];
        blank_row_emit<K, VEC4>(p, r.t + 1, r.b, r.L, v, r.x1, occ, gam, nxt_l, tb);
    } else {
        float v[K];                                          // the sample's last row: beta' is the entry condition
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int s = s0 + k;
            v[k] = (s == n - 1 || s == n - 2) ? a0[k] : kNegB;
        }
        blank_row_emit<K, VEC4>(p, r.t, r.b, r.L, v, r.x0, occ, gam, nxt_l, tb);
        if (r.exist1) blank_row_fill<VEC4>(p, r.t + 1, r.b, 0.f);
    }
}

// wave `first` of `stride` takes pairs first, first+stride, ...; three pairs (six rows) in flight
template <int K, bool VEC4>
__device__ __forceinline__ void blank_grad_pairs(const BlankParams &p, int first, int stride, int total, float *occ, float *gam, float *stage)
{
    int idx = first;
    if (idx >= total) return;
    BlankTables<K> tb;
    ChainsSeen ps;
    BlankPair<K> r0, r1, r2;
    constexpr int NB = 3;
    // (loads are issued for indices past the end as well -- they name no pair and load clamped rows: see blank_pair_load
    // on why the number of loads between a pair and its use must be the same on every path)
    blank_pair_load<K, VEC4>(p, idx, r0, ps);
    blank_pair_load<K, VEC4>(p, idx + stride, r1, ps);
#define CTC_TURN(Q, CUR, NEXT)                                                                                 \
    blank_pair_load<K, VEC4>(p, idx + ((Q) + NB - 1) * stride, NEXT, ps);                                      \
    if (idx + (Q)*stride < total) blank_pair_finish<K, VEC4>(p, CUR, occ, gam, stage, tb);
    for (; idx < total; idx += NB * stride) {
        CTC_TURN(0, r0, r2)
        CTC_TURN(1, r1, r0)
        CTC_TURN(2, r2, r1)
    }
#undef CTC_TURN
}

// VEC4: C % 4 == 0 and 16-byte aligned rows -> the dense part moves float4 per lane (4x fewer
// memory instructions).  Rows are double-buffered: the loads of a wave's NEXT row (lattice
// triples + the whole log-prob row, kMaxV4 float4 per lane cover C <= 1024) are in flight while
// the current row is reduced and written.  Wave `first` of `stride` takes rows first, first+stride, ...
template <int K, bool VEC4, bool SYNC>
__device__ __forceinline__ void blank_grad_rows(const BlankParams &p, int first, int stride, int total_rows, float *occ, float *gam)
{
    int idx = first;
    if (idx >= total_rows) return;
    BlankTables<K> tb;
    ChainsSeen ps;
    if (!SYNC) {
        BlankRow<K> ra, rb;
        blank_row_load<K, VEC4, SYNC>(p, idx, ra, ps);
        for (; idx < total_rows; idx += 2 * stride) {        // (loads past the end name no row: see blank_row_load)
            blank_row_load<K, VEC4, SYNC>(p, idx + stride, rb, ps);
            blank_row_finish<K, VEC4, SYNC>(p, ra, occ, gam, tb);
            blank_row_load<K, VEC4, SYNC>(p, idx + 2 * stride, ra, ps);
            if (idx + stride < total_rows) blank_row_finish<K, VEC4, SYNC>(p, rb, occ, gam, tb);
        }
    } else {
        // fused launch: two waves per SIMD instead of eight, so each keeps six rows in flight (four: +3 %).
        // (Holding the workers back to one row while chains are still running, to leave HBM to the loaders,
        // was slower: 559 against 536 us at config 5.)
        BlankRow<K> r0, r1, r2, r3, r4, r5;
        constexpr int NB = 6;                                // rows in flight
        blank_row_load<K, VEC4, SYNC>(p, idx, r0, ps);
        blank_row_load<K, VEC4, SYNC>(p, idx + stride, r1, ps);
        blank_row_load<K, VEC4, SYNC>(p, idx + 2 * stride, r2, ps);
        blank_row_load<K, VEC4, SYNC>(p, idx + 3 * stride, r3, ps);
        blank_row_load<K, VEC4, SYNC>(p, idx + 4 * stride, r4, ps);
#define CTC_TURN(Q, CUR, NEXT)                                                                                 \
    blank_row_load<K, VEC4, SYNC>(p, idx + ((Q) + NB - 1) * stride, NEXT, ps);                                 \
    if (idx + (Q)*stride < total_rows) blank_row_finish<K, VEC4, SYNC>(p, CUR, occ, gam, tb);
        for (; idx < total_rows; idx += NB * stride) {
            CTC_TURN(0, r0, r5)
            CTC_TURN(1, r1, r0)
            CTC_TURN(2, r2, r1)
            CTC_TURN(3, r3, r2)
            CTC_TURN(4, r4, r3)
            CTC_TURN(5, r5, r4)
        }
#undef CTC_TURN
    }
}

template <int K, bool VEC4>
__global__ __launch_bounds__(kGradWaves * kWave) void blank_grad_kernel(BlankParams p, int total_rows)
{
    extern __shared__ float4 s_buf4[];                       // per wave: occ[C4] + gam[NSP] + nxt[NSP]
    const int w = wave_id(), lane = lane_id();
    const int C4 = (p.C + 3) & ~3;
    float *occ = reinterpret_cast<float *>(s_buf4) + (size_t)w * (C4 + 2 * p.NSP);
    float *gam = occ + C4;
    for (int c = lane; c < C4; c += kWave) occ[c] = 0.f;
    blank_grad_rows<K, VEC4, false>(p, blockIdx.x * kGradWaves + w, gridDim.x * kGradWaves, total_rows, occ, gam);
}

// ---- fused schedule: the launch -----------------------------------------------------------------
template <int K, bool VEC4, bool POOL = false>
__global__ __launch_bounds__(kFusedThreads) void blank_fused_kernel(BlankParams p)
{
    extern __shared__ float4 s_buf4[];
    const int w = wave_id(), lane = lane_id();
    if ((int)blockIdx.x < p.B) {                             // a sample's workgroup (first in dispatch order)
        const int b = blockIdx.x;
        const FusedLds f = fused_lds<K>(p, s_buf4);
        if (threadIdx.x < 16) f.flags[threadIdx.x] = 0;      // (kFusedLdsHead = 64 bytes of flags)
        __syncthreads();
        int Tb, L;
        const bool ok = blank_sample_ok(p, b, Tb, L);
        const bool run = ok && Tb > 0;
        float a[K];
        static_assert(kFusedWaves == 2 + 2 * kLoaders, "two chains and their loaders");
        // waves 0 / 1: the chains; waves 2, 4, 6 / 3, 5, 7: the loaders of the alpha / beta direction
        const int dir = w & 1, role = w >> 1;                // role 0: chain, 1..kLoaders: loader role-1
        constexpr bool kPool = VEC4 && POOL && kPoolGather;  // the worker pool gathers the emission rows (see kPoolGather)
        constexpr int NL = kPool ? kPoolLoaders : kLoaders;  // data loader waves per direction
        if (role == 0) {
            __builtin_amdgcn_s_setprio(3);
            if (dir == 0) {
                if (run) blank_chain_fused<K, true, NL>(p, b, Tb, L, a, f, VEC4 && kGatherOnce && !kPool);
                blank_publish<K>(p, b, ok, Tb, L, a);
            } else if (run) {
                blank_chain_fused<K, false, NL>(p, b, Tb, L, a, f, VEC4 && kGatherOnce && !kPool);
            }
        } else if (run) {
            const int j = role - 1;
            if (kPool) {
                if (j < kPoolLoaders) blank_pool_loader<K>(p, b, Tb, L, dir, j, f);
                else blank_pool_scout(p, b, Tb, f.flags + 12 + dir);
            } else if (VEC4) {
                float *stage = f.ring(2) + (kLoaders * dir + j) * 4 * kWave * kMaxV4;
                blank_loader_rows<K>(p, b, Tb, L, dir, j, f, stage);
            } else {
                blank_loader<K>(p, b, Tb, L, dir, j, f);
            }
        }
        return;
    }
    // a row worker: per wave occ[C4] + gam[NSP] + nxt[NSP] (+ a staged log_probs row when it works on row pairs)
    const int C4 = (p.C + 3) & ~3;
    const int per_wave = C4 + 2 * p.NSP + (kHalfLattice && VEC4 ? 4 * kWave * kMaxV4 : 0);
    float *occ = reinterpret_cast<float *>(s_buf4) + (size_t)w * per_wave;
    float *gam = occ + C4;
    for (int c = lane; c < C4; c += kWave) occ[c] = 0.f;
    const int wid = ((int)blockIdx.x - p.B) * kFusedWaves + w, nw = ((int)gridDim.x - p.B) * kFusedWaves;
    if (wid == 0) bstamp(p, 0);
    if (wid == nw - 1) bstamp(p, 15);
    // rows exist from distance (T_b - 1)/2 on (the middle of the sample): start at the smallest one of the batch
    // (row pairs: the same in units of pairs)
    int m0 = p.T;
    for (int bb = lane; bb < p.B; bb += kWave) {
        const int len = kHalfLattice ? (p.meta[bb].x + 1) >> 1 : p.meta[bb].x;
        m0 = min(m0, (max(len, 1) - 1) >> 1);
    }
#pragma unroll
    for (int sh = 1; sh < kWave; sh <<= 1) m0 = min(m0, __shfl_xor(m0, sh));
    m0 = __builtin_amdgcn_readfirstlane(m0);
    if (VEC4 && POOL && kPoolGather) {                       // the pool's first job: the emission rows, in the chains' order
        blank_pool_gather<K>(p, wid, nw, kHalfLattice ? gam + 2 * p.NSP : occ);
        if (!kHalfLattice)
            for (int c = lane; c < C4; c += kWave) occ[c] = 0.f;
        if (wid == 0) bstamp(p, 11);
    }
    if (kHalfLattice) blank_grad_pairs<K, VEC4>(p, m0 * 2 * p.B + wid, nw, 2 * ((p.T + 1) >> 1) * p.B, occ, gam, gam + 2 * p.NSP);
    else blank_grad_rows<K, VEC4, true>(p, m0 * 2 * p.B + wid, nw, 2 * p.T * p.B, occ, gam);
    if (wid == 0) bstamp(p, 12);
    if (wid == nw - 1) bstamp(p, 13);
    if (wid == nw / 2) bstamp(p, 14);
}

// workgroups of blank_fused_kernel that are resident at once on this device (at most one per CU), 0 when unknown
template <int K, bool VEC4>
static int fused_capacity(size_t lds)
{
    // per device (and per instantiation): the last LDS size asked about and its answer, packed into one
    // atomic word so that concurrent host threads see a consistent pair
    static std::atomic<unsigned long long> cache[kMaxDevices];
    const int dev = current_device();
    const unsigned long long key = (unsigned long long)lds << 16;
    if (dev >= 0) {
        const unsigned long long c = cache[dev].load(std::memory_order_acquire);
        if (c != 0 && (c & ~0xffffull) == key) return (int)(c & 0xffff) - 1;
    }
    int per_cu = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(blank_fused_kernel<K, VEC4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blank_fused_kernel<K, VEC4>, kFusedThreads, lds) != hipSuccess)
        per_cu = 0;
    const int cap = per_cu >= 1 ? device_cus() : 0;
    if (dev >= 0) cache[dev].store(key | (unsigned long long)(cap + 1), std::memory_order_release);
    return cap;
}

// schedule of the long-sequence path: -1 the library's own choice, 1 / 0 force / forbid the persistent launch, 2 force
// it with the worker pool gathering the emission rows
// (ctc_amd_blank_set_schedule; the initial value comes from CTC_AMD_BLANK_FUSED, read once)
static std::atomic<int> g_blank_schedule{-2};
static int blank_schedule()
{
    int v = g_blank_schedule.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = getenv("CTC_AMD_BLANK_FUSED");
        v = e && e[0] == '1' ? 1 : (e && e[0] == '0' ? 0 : -1);
        g_blank_schedule.store(v, std::memory_order_relaxed);
    }
    return v;
}

template <int K>
static int run_blank(BlankParams &p, hipStream_t s)
{
    p.NSP = kWave * K;
    const size_t lattice = (size_t)p.B * p.T * p.NSP;
    float *base = reinterpret_cast<float *>(reinterpret_cast<char *>(p.counter) + 256);
    p.em = base;
    p.al = base + lattice;
    p.be = base + 2 * lattice;
    p.cls = reinterpret_cast<int *>(base + 3 * lattice);
    p.nxt = p.cls + (size_t)p.B * p.NSP;
    p.first = p.nxt + (size_t)p.B * p.NSP;
    p.meta = reinterpret_cast<int2 *>(p.first + (size_t)p.B * p.NSP);
    p.Bp = (p.B + 63) & ~63;
    p.sync = reinterpret_cast<int *>(p.meta + p.Bp);
    p.nsync = blank_sync_ints(p.T, p.B);
    p.nchunk = blank_pool_chunks(p.T);
    p.chunk = p.sync + kSyncHead + 2 * p.Bp * kProgPitch;
    static const int debug = diag_env("CTC_AMD_BLANK_DEBUG");
    p.debug = debug;
    const size_t row_lds = (((p.C + 3) & ~3) + 2 * p.NSP) * sizeof(float);   // a grad wave's occ[] + gam[] + nxt[]
    const bool vec4 = (p.C % 4 == 0) && p.C <= 4 * kWave * kMaxV4 && (p.st % 4 == 0) && (p.sb % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(p.lp) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.grad) % 16 == 0);

    // Persistent launch.  Hard conditions: a gradient is wanted, every workgroup resident with at least as
    // many workers as chains, 32-bit row offsets.  Where it PAYS was measured (tools/blank_sweep.py, T = 1000,
    // us per call persistent / three launches, round 2 -- half of the lattice in memory): B=64 C=1000 S=100 220/296,
    // 64x500x100 180/238, 32x1000x100 170/201, 48x640x100 184/229, 64x800x200 289/524, 64x200x100 184/203,
    // 96x1000x100 378/432, 128x1000x100 436/521, 24x1000x100 174/189 -- but 16x1000x100 169/164 (too little
    // bandwidth work per step), 64x1000x30 192/194 and 64x400x30 138/140 (two states per lane: a tie), 32x400x30
    // 132/111, 32x2000x50 352/285 (rows too wide for the float4 loaders); over T at 64x1000x100: 51.5/51.8 at 128,
    // 74/82 at 256, 119/146 at 512.  ctc_amd_blank_set_schedule(1 / 0) forces / forbids it (tests, measurements).
    const int schedule = blank_schedule();
    const bool forced = schedule >= 1, forbidden = schedule == 0, forced_pool = schedule == 2;
    if (p.grad && !forbidden && p.T >= kFusedMinT && (int64_t)p.T * p.NSP * 4 < kPastLattice &&
        (int64_t)2 * p.T * p.B + 4096 < ((int64_t)1 << 31)) {
        // more than half of a CU's LDS per workgroup: one workgroup per CU, the chains share their SIMDs with nobody
        size_t lds = kFusedLdsHead + 2 * (size_t)(kRingRows / K) * p.NSP * sizeof(float) +
                     (vec4 ? 2 * kLoaders * (size_t)(4 * kWave * kMaxV4) * sizeof(float) : 0);   // + a row per loader
        const size_t worker_lds = row_lds + (kHalfLattice && vec4 ? (size_t)4 * kWave * kMaxV4 * sizeof(float) : 0);
        if (lds < kFusedWaves * worker_lds) lds = kFusedWaves * worker_lds;
        if (lds < kMaxLds / 2 + 1024) lds = kMaxLds / 2 + 1024;
        const int cap = lds <= kMaxLds ? (vec4 ? fused_capacity<K, true>(lds) : fused_capacity<K, false>(lds)) : 0;
        const bool pays = K >= 4 && vec4 && p.T >= 2 * kFusedMinT && 11 * p.B >= cap && 2 * p.B <= cap &&
                          (int64_t)p.B * p.C >= 16384;
        if (cap >= 2 * p.B && cap - p.B >= 32 && (pays || forced)) {
            int rc = launch<blank_tables_kernel>(dim3(p.B), dim3(256), p.NSP * sizeof(int), s, p);
            if (rc) return rc;
            const dim3 grid(cap), block(kFusedThreads);
            // The worker pool gathers the emission rows (kPoolGather) where the pool is large against the samples and the
            // sequences long: tools/blank_sweep.py, T = 1000, us per call pool / loaders -- B=64 C=1000 S=100 209 / 220
            // (T=2000: 384 / 432), but 32x1000x100 175 / 170, 48x640x100 186 / 184 (ties), 96x1000x100 410 / 378,
            // 128x1000x100 489 / 436 (too few workers per sample), 64x800x200 362 / 289 (eight states per lane), and
            // T=512 123 / 119, T=256 89 / 74 (the chains' first steps wait for the pool's first rows).
            // (ctc_amd_blank_set_schedule(2) forces the pool gather on every float4 shape: tests)
            const bool pool = kPoolGather && vec4 &&
                              (forced_pool || (K == 4 && p.T >= 800 && 4 * p.B >= cap && 5 * p.B <= 2 * (cap - p.B)));
            if (pool) return launch<blank_fused_kernel<K, true, true>>(grid, block, lds, s, p);
            if (vec4) return launch<blank_fused_kernel<K, true>>(grid, block, lds, s, p);
            return launch<blank_fused_kernel<K, false>>(grid, block, lds, s, p);
        }
    }

    const int rows_per_block = 8;
    const dim3 ggrid((p.T + rows_per_block - 1) / rows_per_block, p.B);
    int rc = launch<blank_gather_kernel>(ggrid, dim3(256), p.NSP * sizeof(int), s, p, rows_per_block);
    if (rc) return rc;
    rc = launch<blank_chain_kernel<K>>(dim3(p.B), dim3(128), 0, s, p);
    if (rc || !p.grad) return rc;
    const int total = p.T * p.B;
    int blocks = (total + kGradWaves - 1) / kGradWaves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = kGradWaves * row_lds;
    if (vec4) return launch<blank_grad_kernel<K, true>>(dim3(blocks), dim3(kGradWaves * kWave), lds, s, p, total);
    return launch<blank_grad_kernel<K, false>>(dim3(blocks), dim3(kGradWaves * kWave), lds, s, p, total);
}

}  // namespace ctc

using namespace ctc;

extern "C" int ctc_amd_blank_loss_grad(const float *log_probs, int64_t stride_t, int64_t stride_b,
                                       const void *targets, int targets_i64,
                                       const int64_t *in_len, const int64_t *tgt_len,
                                       int T, int B, int C, int S, int blank,
                                       float loss_scale, float grad_scale,
                                       float *nll, float *loss, float *grad,
                                       void *workspace, void *stream)
{
    if (!log_probs || !targets || !in_len || !tgt_len || !nll || !loss || !workspace) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || C < 1 || S < 1 || blank < 0 || blank >= C) return CTC_AMD_ERR_BAD_ARGUMENT;
    const int ns = 2 * S + 1;
    if (ns > kWave * 8) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;                 // S <= 255
    if ((size_t)kGradWaves * (C + 4 + 2 * kWave * 8) * sizeof(float) > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    BlankParams p;
    p.lp = log_probs; p.st = stride_t; p.sb = stride_b;
    p.tgt = targets; p.tgt64 = targets_i64;
    p.in_len = in_len; p.tgt_len = tgt_len;
    p.T = T; p.B = B; p.C = C; p.S = S; p.blank = blank;
    p.loss_scale = loss_scale; p.grad_scale = grad_scale;
    p.nll = nll; p.loss = loss; p.grad = grad;
    p.counter = static_cast<unsigned *>(workspace);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ns <= kWave * 2) return run_blank<2>(p, s);
    if (ns <= kWave * 4) return run_blank<4>(p, s);
    return run_blank<8>(p, s);
}

extern "C" int ctc_amd_blank_set_schedule(int mode)
{
    if (mode < -1 || mode > 2) return CTC_AMD_ERR_BAD_ARGUMENT;
    g_blank_schedule.store(mode, std::memory_order_relaxed);
    return 0;
}
