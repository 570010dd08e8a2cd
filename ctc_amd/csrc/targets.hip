// Target construction on the device: the reference's dataset preparation turns the multi-hot
// label rows of a clip into the label SEQUENCE the CTC losses consume -- every distinct non-empty row once,
// in order of first appearance, the rest of the [S, C] block filled with -1, plus the sequence length
// (datasets/charades_ctc_next_pred.py:646-651 / :503-505 the row code, :663-671 / :523-531 the walk,
// :676-678 the -1 padding).
//
// HOW ROWS ARE COMPARED is the reference's own arithmetic: the code of a row is accumulated into an IntTensor,
//     code[t] += row[t, o] * 2**o        (int32; torch wraps the Python integer 2**o to 32 bits)
// so class 31 is the sign bit, classes 32..63 contribute nothing, and at o >= 64 torch raises OverflowError
// ("int too big to convert") whatever the row holds.  A row enters when its code is `not in` the array of kept
// codes, which starts as zeros and is indexed by t: code 0 is always "already there".  At the reference's
// default class counts (opts.py:60-61: 38 object classes, 33 verb classes) rows that differ only in classes
// >= 32 therefore collide and rows made only of such classes never enter.  That is the DEFAULT here
// (exact_rows = 0, bit-exact with the reference at every C <= 64; C > 64 returns CTC_AMD_ERR_CODE_OVERFLOW
// where the reference raises).  exact_rows = 1 compares whole rows (all C bits, any C) instead.
//
// One 256-thread workgroup per clip.  Rows become signatures in LDS (reference mode: the one wrapped 32-bit
// code; exact mode: one word per 32 classes); wave 0 then walks the rows in order and keeps a row when no kept
// signature equals it (kept signatures sit one per lane, 64 at a time); all waves copy the kept rows out.
#include "launch.hpp"

namespace ctc {

template <bool EXACT>
__global__ __launch_bounds__(256) void dedup_rows_kernel(const int32_t *rows, int S, int C, int W, int32_t *out, int64_t *length)
{
    extern __shared__ unsigned dd_smem[];
    unsigned *sig = dd_smem;                                 // [S][W]
    int *kept = reinterpret_cast<int *>(sig + (size_t)S * W);   // [S] row index of the j-th kept row; [S] = count
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t *src = rows + (size_t)b * S * C;
    for (int i = tid; i < S * W; i += 256) {
        const int t = i / W, q = i - t * W;
        unsigned bits = 0;
        for (int k = 0; k < 32; ++k) {
            const int c = 32 * q + k;
            if (c < C) {
                const unsigned v = (unsigned)src[(size_t)t * C + c];
                // reference: the int32 product row * 2**o summed with wrap-around (any integer row value);
                // exact: which classes are set
                bits = EXACT ? (bits | (v != 0 ? 1u << k : 0u)) : bits + (v << k);
            }
        }
        sig[i] = bits;
    }
    __syncthreads();
    if (w == 0) {
        int n = 0;
        for (int t = 0; t < S; ++t) {
            const unsigned *st = sig + (size_t)t * W;
            bool any = false;
            for (int q = 0; q < W; ++q) any |= st[q] != 0;   // (uniform: every lane reads the same words)
            bool seen = !any;                                // code 0 is "already there": never kept
            for (int j0 = 0; j0 < n && !seen; j0 += 64) {
                bool eq = j0 + lane < n;
                if (eq) {
                    const unsigned *sk = sig + (size_t)kept[j0 + lane] * W;
                    for (int q = 0; q < W; ++q) eq &= sk[q] == st[q];
                }
                seen = __builtin_amdgcn_ballot_w64(eq) != 0;
            }
            if (!seen) {
                if (lane == 0) kept[n] = t;
                ++n;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the new entry is read by other lanes next round
            }
        }
        if (lane == 0) {
            kept[S] = n;
            length[b] = n;
        }
    }
    __syncthreads();
    const int n = kept[S];
    int32_t *dst = out + (size_t)b * S * C;
    for (int i = tid; i < S * C; i += 256) {
        const int j = i / C, c = i - j * C;
        dst[i] = j < n ? src[(size_t)kept[j] * C + c] : -1;
    }
}

}  // namespace ctc

using namespace ctc;

extern "C" int ctc_amd_dedup_multihot_targets(const int32_t *rows, int B, int S, int C, int exact_rows,
                                              int32_t *out, int64_t *length, void *stream)
{
    if (!rows || !out || !length) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (B < 1 || S < 1 || C < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (!exact_rows && C > 64) return CTC_AMD_ERR_CODE_OVERFLOW;   // the reference's 2**o raises OverflowError at o = 64
    const int W = exact_rows ? (C + 31) / 32 : 1;
    const size_t smem = ((size_t)S * W + S + 1) * sizeof(unsigned);
    if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    return exact_rows ? launch<dedup_rows_kernel<true>>(dim3(B), dim3(256), smem, s, rows, S, C, W, out, length)
                      : launch<dedup_rows_kernel<false>>(dim3(B), dim3(256), smem, s, rows, S, C, W, out, length);
}
