// Shared device helpers for the gfx950 CTC kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Compile-time experiment switches (-DCTC_X_*: ablations and measured-and-dropped variants) change
// what the kernels compute or skip.  They exist for A/B builds only (tools/build_variant.sh) and need
// -DCTC_AMD_EXPERIMENTS beside them; ctc_amd/build.py refuses both for the product library's name, so a
// stray -D in the environment cannot ship a library that leaves out the batch sum or the status word.
#if !defined(CTC_AMD_EXPERIMENTS) &&                                                                            \
    (defined(CTC_X_NOSTATUS) || defined(CTC_X_NOREDUCE) || defined(CTC_X_NORETURN) || defined(CTC_X_NOCHAIN) || \
     defined(CTC_X_NOPRIO) || defined(CTC_X_NOREAD) || defined(CTC_X_NOWRITE) || defined(CTC_X_FULL_LATTICE) || \
     defined(CTC_X_GATHER_ONCE) || defined(CTC_X_WMASK) || defined(CTC_X_NOPROG) || defined(CTC_X_NORENORM) || \
     defined(CTC_X_ROWDPP) || defined(CTC_X_SCALAR_STEP) || defined(CTC_X_NO_POOL_GATHER) || defined(CTC_X_FLOW_NO_GPRIO) || defined(CTC_X_FLOW_NOEXP) || defined(CTC_X_FLOW_NO_DELEGATE) || defined(CTC_X_FLOW_DELEGATE_OWN) || \
     defined(CTC_FLOW_TILE_WAVES))
#error "CTC_X_* experiment switches need -DCTC_AMD_EXPERIMENTS (A/B builds only, never the product library)"
#endif

namespace ctc {

constexpr int kWave = 64;
// zero_padding sentinel of the reference (NoBlankCTC.py:25): keeping it (instead of
// -inf) keeps every intermediate finite and the float32 roundings identical.
constexpr float kNeg = -10000000000000.0f;
constexpr float kInfeasible = 1.0e12f;   // nll above this <=> no alignment exists

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
// wave-uniform by construction: keep it in an SGPR so row indices / LDS row addresses are scalar
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

// The gradient is written once and not read again by this launch: a non-temporal store keeps
// it from piling up as dirty lines in L2 that the end-of-kernel write-back then has to drain
// (measured at config 2: 21.3 -> 18.9 us per launch).
__device__ __forceinline__ void stream_store(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void stream_store(float4 *p, float4 v)
{
    typedef float native4 __attribute__((ext_vector_type(4)));
    native4 n = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(n, reinterpret_cast<native4 *>(p));
}

// Write-through store (sc1): the bytes leave the XCD's L2 as they are written.  For a launch whose
// whole gradient fits in the L2s (B <= #CUs: 24 MB at config 2) a non-temporal store still parks its
// lines there and the end of the kernel waits for their write-back; tools/micro/rw_phase.hip, B = 256,
// T = 150, C = 158, read everything then write everything: 9.4 us with 8-byte non-temporal stores,
// 7.6 us with 8-byte write-through stores (16-byte: 7.8 / 5.8 us).  At B = 2048 (reads and writes of
// different samples mixing, gradient far beyond L2) non-temporal wins (54 against 79 us).
typedef float wt_f2 __attribute__((ext_vector_type(2)));
typedef float wt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void through_store(wt_f2 *p, wt_f2 v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// (A vector-memory store of MORE than 64 bits reads its data registers over several cycles: a VALU instruction that
// overwrites them must keep 2 wait states behind it on gfx940+ -- LLVM's hazard recognizer inserts them for stores it
// knows, and knows nothing of an inline-asm one.  Without the s_nop a packed multiply-add scheduled right behind the store
// changed the last component of the last lanes' data: wrong gradient values in columns 51, 55, 59, 63 of a row, found
// when the order of the tile reads and the stores in R16Row::store_grad changed; the same hazard is behind the
// "address-select" scatter variant of round 4 that gave wrong label-smoothed gradients.)
__device__ __forceinline__ void through_store(wt_f4 *p, wt_f4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// gradient store of a launch: write-through while logits + gradient fit the memory-side cache, non-temporal
// beyond it (r16 kernel, T = 150, C = 158, us per launch, write-through / non-temporal: B = 512 25.9 / 30.6,
// 1024 49.7 / 54.6, 1536 83.9 / 78.2, 2048 133.5 / 101.3)
template <bool NT, typename V>
__device__ __forceinline__ void grad_store(V *p, V v)
{
    if (NT) {
        typedef V __attribute__((aligned(8))) VA;
        __builtin_nontemporal_store(v, reinterpret_cast<VA *>(p));
    } else {
        through_store(p, v);
    }
}

// Workspace word 2 (bytes [8,12)): STATUS.  A bounded in-kernel wait that ran out ORs its bit in here
// (and poisons the outputs it could not produce with NaN); nothing in the kernels ever clears it --
// ctc_amd_workspace_status() reads / clears it from the host.  Never observed outside fault-injection
// builds: the hand-offs cannot break unless a workgroup is starved for ~1 s.
constexpr unsigned kStatusNoblankStarved = 1u, kStatusBinaryStarved = 2u, kStatusBlankStarved = 4u;
__device__ __forceinline__ void raise_status(unsigned *counter, unsigned bit)
{
#ifdef CTC_X_NOSTATUS
    return;
#endif
    if (lane_id() == 0) __hip_atomic_fetch_or(counter + 2, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ARRIVALS -- how many workgroups of the launch in flight have started: read by ctc_amd_collective_gate (api.hip), which
// holds a collective back until the loss launch has filled the chip.  Counting is OFF until a gate
// has been used on the workspace (word 11, bytes [44,48): set by the gate kernel, read here through the scalar cache): one
// atomic per workgroup on ONE word at kernel entry -- 256 of them at once, served at ~90 per us by the memory side -- delays
// every workgroup's first row loads (measured at config 2: 12.3 -> 13.9 us per launch).  When on, the count is sharded
// like the batch sum (sample b adds to the free upper half of shard slot b % 16: bytes [256 + 16 s + 8, +4)), issued by a
// wave that has no other vector-memory operation to wait for (a chain / scan wave); the workgroup that completes the
// batch sum (the last to finish, so every workgroup has arrived) puts the sixteen words back to 0.
constexpr int kGateWord = 11;
__device__ __forceinline__ unsigned *arrival_shard(unsigned *counter, int sh) { return counter + 64 + 4 * sh + 2; }
__device__ __forceinline__ bool gate_on(const unsigned *counter)
{
    typedef const __attribute__((address_space(4))) unsigned const_u32;
    return *(const_u32 *)(uintptr_t)(counter + kGateWord) != 0;
}
__device__ __forceinline__ void note_arrival(unsigned *counter, int b)
{
    if (!gate_on(counter)) return;                           // (wave-uniform: a scalar load)
    __hip_atomic_fetch_add(arrival_shard(counter, b & 15), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void reset_arrivals(unsigned *counter)
{
    // unconditional: a gate that switches the counting on WHILE a launch is in flight lets some of its workgroups count
    // and the finisher (which may have read "off") must still leave the words at 0 for the next gate -- sixteen relaxed
    // stores by one lane of the launch's last workgroup
    for (int sh = 0; sh < 16; ++sh) __hip_atomic_store(arrival_shard(counter, sh), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Diagnostics (phase stamps, early exits) compile to nothing in the product library.
#ifdef CTC_AMD_DIAGNOSTICS
#define CTC_DIAG(p) ((p).stop)
#else
#define CTC_DIAG(p) 0
#endif

// Publish / consume points of LDS hand-offs between waves of a workgroup.  The hardware
// completes a wave's LDS operations in order, so no wait is needed -- but the COMPILER must not
// move a row access across the counter access (float rows and int counters do not alias for
// it).  Zero instructions.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// A value the compiler knows nothing about any more (volatile: never hoisted, merged or duplicated): what is computed
// from it inside a loop stays inside the loop.
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an L2).  Rows of
// neighbouring samples share 128-byte lines (a row is C*4 bytes, rarely a multiple of 128),
// so consecutive samples are mapped onto the SAME XCD: the straddling lines are then fetched
// from HBM once per XCD instead of once per neighbour.  Bijective for any B; speed only.
__device__ __forceinline__ int xcd_sample(int bid, int B)
{
    const int x = bid & 7, k = bid >> 3, q = B >> 3, r = B & 7;
    return x * q + (x < r ? x : r) + k;
}

// lane i <- lane i-1 (lane 0 keeps `fill`): DPP wave_shr:1, one VALU op, no LDS.
__device__ __forceinline__ float wave_shr1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// lane i <- lane i+1 (lane 63 keeps `fill`): DPP wave_shl:1.
__device__ __forceinline__ float wave_shl1(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// DPP building block: lanes selected by `ctrl`/`row_mask` read `v` of their source lane,
// all other lanes get `ident` (so a following op leaves them unchanged).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp(float v, float ident)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        __builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143;

// all-reduce inside each 16-lane DPP row: 4 VALU ops, no LDS round trip
__device__ __forceinline__ float row16_max(float v)
{
    const float ni = -__builtin_inff();
    v = fmaxf(v, dpp<kQuadXor1, 0xf>(v, ni));
    v = fmaxf(v, dpp<kQuadXor2, 0xf>(v, ni));
    v = fmaxf(v, dpp<kRowHalfMirror, 0xf>(v, ni));
    v = fmaxf(v, dpp<kRowMirror, 0xf>(v, ni));
    return v;
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp<kQuadXor1, 0xf>(v, 0.f);
    v += dpp<kQuadXor2, 0xf>(v, 0.f);
    v += dpp<kRowHalfMirror, 0xf>(v, 0.f);
    v += dpp<kRowMirror, 0xf>(v, 0.f);
    return v;
}
// whole-wave reductions: rows combined with row_bcast, result read from lane 63 into an
// SGPR (wave-uniform).  6 DPP ops + 1 readlane.
__device__ __forceinline__ float wave_max(float v)
{
    const float ni = -__builtin_inff();
    v = row16_max(v);
    v = fmaxf(v, dpp<kRowBcast15, 0xa>(v, ni));
    v = fmaxf(v, dpp<kRowBcast31, 0xc>(v, ni));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v)
{
    v = row16_sum(v);
    v += dpp<kRowBcast15, 0xa>(v, 0.f);
    v += dpp<kRowBcast31, 0xc>(v, 0.f);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Four independent whole-wave reductions at once, hand-scheduled: the four chains are
// interleaved so every DPP source was written >= 3 instructions earlier (the VALU->DPP
// hazard needs 2 wait states; hipcc does not pad inside asm), one instruction per DPP step
// (v_max/v_add with the DPP operand fused, no identity fill, no NaN canonicalisation).
// 24 VALU for four reductions, against ~30 (max) per reduction from the generic form.
#define CTC_DPP4(OP, CTRL)                                     \
    OP " %0, %0, %0 " CTRL " bank_mask:0xf\n\t"                \
    OP " %1, %1, %1 " CTRL " bank_mask:0xf\n\t"                \
    OP " %2, %2, %2 " CTRL " bank_mask:0xf\n\t"                \
    OP " %3, %3, %3 " CTRL " bank_mask:0xf\n\t"
#define CTC_REDUCE4(OP)                                        \
    "s_nop 1\n\t"                                              \
    CTC_DPP4(OP, "quad_perm:[1,0,3,2] row_mask:0xf")           \
    CTC_DPP4(OP, "quad_perm:[2,3,0,1] row_mask:0xf")           \
    CTC_DPP4(OP, "row_half_mirror row_mask:0xf")               \
    CTC_DPP4(OP, "row_mirror row_mask:0xf")                    \
    CTC_DPP4(OP, "row_bcast:15 row_mask:0xa")                  \
    CTC_DPP4(OP, "row_bcast:31 row_mask:0xc")                  \
    "s_nop 0"
__device__ __forceinline__ float lane63(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ void wave_max4(float &a, float &b, float &c, float &d)
{
    asm volatile(CTC_REDUCE4("v_max_f32_dpp") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    a = lane63(a); b = lane63(b); c = lane63(c); d = lane63(d);
}
__device__ __forceinline__ void wave_sum4(float &a, float &b, float &c, float &d)
{
    asm volatile(CTC_REDUCE4("v_add_f32_dpp") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    a = lane63(a); b = lane63(b); c = lane63(c); d = lane63(d);
}

// Two independent reductions over the two 32-lane HALVES of the wave (two rows of <= 32 states
// side by side): five DPP steps, two chains interleaved plus one s_nop per step for the
// VALU->DPP hazard.  Half totals end up in lanes 31 and 63; every lane gets its half's value.
#define CTC_DPP2(OP, CTRL)                                     \
    OP " %0, %0, %0 " CTRL " bank_mask:0xf\n\t"                \
    OP " %1, %1, %1 " CTRL " bank_mask:0xf\n\t"                \
    "s_nop 0\n\t"
#define CTC_REDUCE2_HALVES(OP)                                 \
    "s_nop 1\n\t"                                              \
    CTC_DPP2(OP, "quad_perm:[1,0,3,2] row_mask:0xf")           \
    CTC_DPP2(OP, "quad_perm:[2,3,0,1] row_mask:0xf")           \
    CTC_DPP2(OP, "row_half_mirror row_mask:0xf")               \
    CTC_DPP2(OP, "row_mirror row_mask:0xf")                    \
    CTC_DPP2(OP, "row_bcast:15 row_mask:0xa")
// two whole-wave sums at once (six fused DPP steps, the two chains interleaved): 12 VALU + 2 readlanes for both,
// against ~20 VALU per sum from the generic form above
#define CTC_REDUCE2_WAVE(OP)                                   \
    CTC_REDUCE2_HALVES(OP)                                     \
    CTC_DPP2(OP, "row_bcast:31 row_mask:0xc")
__device__ __forceinline__ void wave_sum2(float &a, float &b)
{
    asm volatile(CTC_REDUCE2_WAVE("v_add_f32_dpp") : "+v"(a), "+v"(b));
    a = lane63(a); b = lane63(b);
}
__device__ __forceinline__ float half_pick(float v, bool upper)
{
    const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return upper ? hi : lo;
}
__device__ __forceinline__ void halves_sum2(float &a, float &b, bool upper)
{
    asm volatile(CTC_REDUCE2_HALVES("v_add_f32_dpp") : "+v"(a), "+v"(b));
    a = half_pick(a, upper); b = half_pick(b, upper);
}
__device__ __forceinline__ void halves_max2(float &a, float &b, bool upper)
{
    asm volatile(CTC_REDUCE2_HALVES("v_max_f32_dpp") : "+v"(a), "+v"(b));
    a = half_pick(a, upper); b = half_pick(b, upper);
}

// all-reduce inside aligned groups of G = 16 / 32 / 64 lanes, every lane gets the result
template <bool MAX>
__device__ __forceinline__ float group_reduce(float v, int G)
{
    v = MAX ? row16_max(v) : row16_sum(v);
    if (G >= 32) { const float o = __shfl_xor(v, 16, kWave); v = MAX ? fmaxf(v, o) : v + o; }
    if (G >= 64) { const float o = __shfl_xor(v, 32, kWave); v = MAX ? fmaxf(v, o) : v + o; }
    return v;
}

__device__ __forceinline__ float vmax(float a, float b)
{   // plain v_max_f32: both inputs are VALU results, no NaN canonicalisation wanted (fmaxf costs a v_max x, x, x per input)
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float vmax3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
// natural log for arguments in the normal range (raw v_log_f32, no denormal fix-up)
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * kLn2; }

// _logsumexp over two values (NoBlankCTC.py:16-19): max + log(exp(a-max)+exp(b-max))
// == max + log(1 + exp(-|a-b|)); the log argument lies in [1,2], so the raw
// v_exp_f32 / v_log_f32 pair is exact enough (1 ulp) and needs no range fix-up.
__device__ __forceinline__ float lse2(float a, float b)
{
    const float m = a > b ? a : b;
    const float t = __builtin_amdgcn_exp2f(-fabsf(a - b) * kLog2e);
    return __builtin_fmaf(__builtin_amdgcn_logf(1.0f + t), kLn2, m);
}

// labels arrive as int32 or int64 (little-endian): one branch-free 32-bit load of the
// low word serves both (class indices and the -1 padding fit in 32 bits).
__device__ __forceinline__ int load_label(const void *p, int is64, int64_t i)
{
    return static_cast<const int32_t *>(p)[is64 ? 2 * i : i];
}

// Two int64 values at wave-uniform addresses through the scalar memory path (s_load_dwordx2): loaded through
// constant-address-space pointers, so the COMPILER emits the scalar loads and tracks them (its own
// s_waitcnt lgkmcnt in front of the first use; a hand-written asm load would leave the destination SGPRs
// "defined" while the load is still in flight, free for the register allocator to copy or spill).  A plain
// global load of a uniform address becomes a vector load + readfirstlane instead: a memory round trip in
// front of everything that follows, and every later wait on the row loads degrades to vmcnt(0).
struct ScalarLengths {
    int64_t a, b;
    __device__ __forceinline__ ScalarLengths() : a(0), b(0) {}
    __device__ __forceinline__ ScalarLengths(const int64_t *pa, const int64_t *pb)
    {
        typedef const __attribute__((address_space(4))) int64_t const_i64;
        a = *(const_i64 *)(uintptr_t)pa;
        b = *(const_i64 *)(uintptr_t)pb;
    }
    __device__ __forceinline__ void get(int64_t &va, int64_t &vb) const
    {
        va = a;
        vb = b;
    }
};

// Diagnostics only (CTC_AMD_DEBUG_STOP = -(wave+1)): that wave of workgroup 0 stamps
// (s_memtime, s_memrealtime) pairs into workspace bytes [64,256) at phase boundaries
// (tools/stamps.py reads them).  Never executes in a normal run (p.stop == 0).
// (binary variant, NoBlankBinaryCTC.py:146,:112) p = sigmoid(x) in fp32, then the two clamped logs exactly as nn.BCELoss sees them
__device__ __forceinline__ void bce_logs(float x, float &p, float &lp, float &lq)
{
    p = 1.0f / (1.0f + expf(-x));
    lp = fmaxf(logf(p), -100.0f);
    lq = fmaxf(logf(1.0f - p), -100.0f);
}

template <typename P>
__device__ __forceinline__ void stamp(const P &p, int slot)
{
    if (CTC_DIAG(p) >= 0 || CTC_DIAG(p) <= -100) return;
    if (blockIdx.x == 0 && wave_id() == -CTC_DIAG(p) - 1 && lane_id() == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * slot;
        o[0] = __builtin_amdgcn_s_memtime();
        o[1] = __builtin_amdgcn_s_memrealtime();
    }
}

// Second diagnostic mode (CTC_AMD_DEBUG_STOP = -100 - wave): that wave of workgroup 0 stamps the steps
// of the kernel's SETUP instead (the phase stamps stay silent), same slots, same reader.
template <typename P>
__device__ __forceinline__ void stamp_setup(const P &p, int slot)
{
    if (CTC_DIAG(p) > -100 || CTC_DIAG(p) <= -200) return;
    if (blockIdx.x == 0 && wave_id() == -CTC_DIAG(p) - 100 && lane_id() == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * slot;
        o[0] = __builtin_amdgcn_s_memtime();
        o[1] = __builtin_amdgcn_s_memrealtime();
    }
}

// Third diagnostic mode (CTC_AMD_DEBUG_STOP = base - wave, base = -200, -300, ...): kernels that loop over samples
// stamp one slot per SAMPLE (slot = sample index % 12) at one point of the loop; same slots, same reader.
template <typename P>
__device__ __forceinline__ void stamp_mode(const P &p, int base, int slot)
{
    if (CTC_DIAG(p) > base || CTC_DIAG(p) <= base - 64) return;
    if (blockIdx.x == 0 && wave_id() == base - CTC_DIAG(p) && lane_id() == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(p.counter) + 8 + 2 * (slot % 12);
        o[0] = __builtin_amdgcn_s_memtime();
        o[1] = __builtin_amdgcn_s_memrealtime();
    }
}

// Deterministic batch reduction by the LAST workgroup to finish its nll (in-launch
// arrival ticket).  Caller: exactly one wave per workgroup, after lane 0 has the
// sample's value.  nll is stored write-through (sc1), drained, then the ticket is
// drawn; the workgroup drawing B-1 reads every nll with sc1 loads in a fixed order.
// (two halves, so that a caller can start the write-through store early and draw the ticket later)
__device__ __forceinline__ void publish_value(float value, int b, float *nll)
{
    if (lane_id() == 0) __hip_atomic_store(&nll[b], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename F>
__device__ __forceinline__ void ticket_and_reduce(int B, float *nll, float *loss, float loss_scale, unsigned *counter,
                                                  F per_sample)
{
    int last = 0;
    if (lane_id() == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's nll store has reached L2
        unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (ticket == (unsigned)(B - 1));
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    float s = 0.f;
    for (int i = lane_id(); i < B; i += kWave)
        s += per_sample(__hip_atomic_load(&nll[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), i);
    s = wave_sum(s);
    if (lane_id() == 0) {
        loss[0] = s * loss_scale;
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        reset_arrivals(counter);
    }
}
template <typename F>
__device__ __forceinline__ void publish_and_reduce(float value, int b, int B, float *nll,
                                                   float *loss, float loss_scale,
                                                   unsigned *counter, F per_sample)
{
    publish_value(value, b, nll);
#ifndef CTC_X_NOREDUCE
    ticket_and_reduce(B, nll, loss, loss_scale, counter, per_sample);
#endif
}

// Batch sum of the per-sample values without the three dependent round trips of publish_and_reduce
// (nll store -> acknowledgement -> ticket -> last arriver reads every nll -> loss), which cost the
// kernel its tail through queues full of gradient stores (config 2: 13.2 us with, 12.1 us without).
// Every workgroup makes ONE returning 64-bit atomic add to a packed word
//     [63:52] arrivals   [51:40] arrivals that did not fit   [39:0] sum of value * 2^F (fixed point);
// integer adds commute, so whoever completes a word holds its exact sum: bitwise the same from run to
// run, exact to 2^-(F+1) per sample.  F depends on B alone (every workgroup must use the same): a value
// fits when it is below 2^13 = 8192, so B of them need 13 + ceil(log2 B) integer bits and F = 27 -
// ceil(log2 B) (B = 256: 19 bits, one sample's rounding 9.5e-7; B = 4095: 15 bits; never below the fp32
// resolution of an nll >= 16 at B <= 256).  Atomics on ONE word are served at ~90 per us (256 workgroups
// finishing together wait 2.8 us for each other), so the words are SHARDED: sample b adds to shard
// b % 16, the workgroup that completes a shard adds the shard's word to the top word, and the one that
// completes the top word writes the loss.  The nll output is a write-through store nobody waits for.
// A value that does not fit (value >= 8192, negative, NaN, inf: the infeasible sentinel 1e13, a
// starved hand-off) takes the slow way for ITSELF only -- store, acknowledgement, its index into a
// list, acknowledgement, then the packed add counting it as "did not fit" -- and the finisher adds those
// from memory (in double).  B > 4095: the ticket form.
// Workspace: bytes [16,24) top word, [24,28) list length, [256,512) shard words, [512, 512 + 4 B) the list.
constexpr int kAccIntBits = 13, kAccMaxB = 4095, kAccShards = 16;
__device__ __forceinline__ int acc_frac_bits(int B)          // 40 - 13 - ceil(log2 B), B in [1, 4095]
{
    return 40 - kAccIntBits - (B <= 1 ? 0 : 32 - __builtin_clz((unsigned)(B - 1)));
}
__host__ __device__ inline size_t acc_list_bytes(int B) { return 256 + (((size_t)4 * B + 255) & ~(size_t)255); }
__device__ __forceinline__ void publish_and_reduce_sum(float value, int b, int B, float *nll, float *loss,
                                                       float loss_scale, unsigned *counter)
{
    if (B > kAccMaxB) {
        publish_and_reduce(value, b, B, nll, loss, loss_scale, counter, [](float x, int) { return x; });
        return;
    }
    if (lane_id() != 0) return;
    unsigned long long *top = reinterpret_cast<unsigned long long *>(counter + 4);
    unsigned long long *shards = reinterpret_cast<unsigned long long *>(counter + 64);
    unsigned *nlist = counter + 6, *list = counter + 128;
    __hip_atomic_store(&nll[b], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long add = 1ull << 52;
    const int frac = acc_frac_bits(B);
    if (value >= 0.f && value < (float)(1 << kAccIntBits)) {
        add += (unsigned long long)__double2ull_rn((double)value * (double)(1u << frac));
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this value is in memory ...
        const unsigned slot = __hip_atomic_fetch_add(nlist, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&list[slot], (unsigned)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and on the list before it is counted
        add += 1ull << 40;
    }
    const int sh = b % kAccShards;
    const int members = (B - sh + kAccShards - 1) / kAccShards;          // samples b' < B with b' % 16 == sh
    unsigned long long *word = shards + 2 * sh;                          // 16-byte pitch
    unsigned long long old = __hip_atomic_fetch_add(word, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CTC_X_NORETURN
    return;
#endif
    if ((old >> 52) != (unsigned long long)(members - 1)) return;
    add += old;                                                          // the shard, complete
    __hip_atomic_store(word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __hip_atomic_fetch_add(top, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 52) + (add >> 52) != (unsigned long long)B) return;
    const unsigned long long tot = old + add;
    double s = (double)(tot & ((1ull << 40) - 1)) * (1.0 / (double)(1u << frac));
    const unsigned nsp = (unsigned)(tot >> 40) & 0xfffu;
    for (unsigned i = 0; i < nsp; ++i) {
        const unsigned bb = __hip_atomic_load(&list[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s += (double)__hip_atomic_load(&nll[bb < (unsigned)B ? bb : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    loss[0] = (float)(s * (double)loss_scale);
    reset_arrivals(counter);
    __hip_atomic_store(top, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(nlist, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace ctc
