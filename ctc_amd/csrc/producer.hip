// The step immediately upstream of the loss: the reference's LSTM_cell.forward (LSTM.py:39-51) runs
//     for time in range(temporal):  v = self.v(feat[time]);  v_hsn, v_csn = self.v_cell(v, (v_hsn, v_csn));
//                                   v_series[time] = v_hsn
// i.e. one torch.nn.LSTMCell step per frame whose hidden state IS the logits row the CTC losses read.  This file
// is that step as ONE launch: both gate products, the cell update, and the hidden state written straight into
// v_series[time] -- in the row layout the loss kernels want (unit stride over classes, any row pitch; pad columns
// behind the last class filled with a value of the caller's choice: a pitch of C + 1 with -1e30 there turns an
// odd class count (the reference's 33) into the even, 8-byte aligned rows of the four-rows-per-wave loss kernel
// without changing a single loss or gradient value -- softmax gives that column exactly 0).
//
// torch.nn.LSTMCell semantics:  gates = x W_ih^T + b_ih + h W_hh^T + b_hh, chunks (i, f, g, o) of H rows each;
//     i, f, o = sigmoid, g = tanh;   c' = f c + i g;   h' = o tanh(c').
//
// Small and latency-bound at the reference's sizes (B = 10, H = 33: 17 kFLOP per step) and still small at the
// benchmark's (B = 256, H = 158: 0.1 GFLOP): one workgroup per kLstmSamples samples stages [x | h] in LDS, every
// thread owns gate rows r = tid, tid + 256, ... of [W_ih | W_hh] and walks them once for all staged samples (the
// weights, <= 0.8 MB, stay in L2; the staged inputs are LDS broadcasts), the pre-activations meet in LDS and one
// thread per hidden unit finishes the cell.  fp32 fma chains in k order (W_ih part, then W_hh part, then the biases).
#include "common.hpp"
#include "launch.hpp"

namespace ctc {

constexpr int kLstmSamples = 8, kLstmThreads = 256;

struct LstmParams {
    const float *x, *h, *c, *w_ih, *w_hh, *b_ih, *b_hh;
    int B, I, H;
    float *h_out, *c_out, *gates;                            // gates: optional [B][4H] activations (i, f, g, o)
    float *series;                                           // optional: row b at series + b * series_stride_b
    int64_t series_stride_b;
    int series_cols;                                         // columns [H, series_cols) of a row get pad_value
    float pad_value;
};

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

__global__ __launch_bounds__(kLstmThreads) void lstm_cell_step_kernel(LstmParams p)
{
    extern __shared__ float lstm_smem[];
    const int K = p.I + p.H, G = 4 * p.H;
    float *xh = lstm_smem;                                   // [kLstmSamples][K]
    float *pre = xh + kLstmSamples * K;                      // [kLstmSamples][G]
    const int tid = threadIdx.x, b0 = blockIdx.x * kLstmSamples;
    const int ns = min(kLstmSamples, p.B - b0);
    for (int i = tid; i < kLstmSamples * K; i += kLstmThreads) {
        const int s = i / K, k = i - s * K;
        float v = 0.f;
        if (s < ns) v = k < p.I ? p.x[(size_t)(b0 + s) * p.I + k] : p.h[(size_t)(b0 + s) * p.H + (k - p.I)];
        xh[i] = v;
    }
    __syncthreads();
    for (int r = tid; r < G; r += kLstmThreads) {
        float acc[kLstmSamples];
#pragma unroll
        for (int s = 0; s < kLstmSamples; ++s) acc[s] = 0.f;
        const float *wi = p.w_ih + (size_t)r * p.I, *wh = p.w_hh + (size_t)r * p.H;
#pragma unroll 4
        for (int k = 0; k < p.I; ++k) {
            const float w = wi[k];
#pragma unroll
            for (int s = 0; s < kLstmSamples; ++s) acc[s] = __builtin_fmaf(w, xh[s * K + k], acc[s]);
        }
#pragma unroll 4
        for (int k = 0; k < p.H; ++k) {
            const float w = wh[k];
#pragma unroll
            for (int s = 0; s < kLstmSamples; ++s) acc[s] = __builtin_fmaf(w, xh[s * K + p.I + k], acc[s]);
        }
        const float bias = p.b_ih[r] + p.b_hh[r];
#pragma unroll
        for (int s = 0; s < kLstmSamples; ++s) pre[s * G + r] = acc[s] + bias;
    }
    __syncthreads();
    for (int i = tid; i < ns * p.H; i += kLstmThreads) {
        const int s = i / p.H, j = i - s * p.H, b = b0 + s;
        const float *g4 = pre + s * G;
        const float gi = sigmoid_f(g4[j]), gf = sigmoid_f(g4[p.H + j]), gg = tanhf(g4[2 * p.H + j]), go = sigmoid_f(g4[3 * p.H + j]);
        const float cn = __builtin_fmaf(gf, p.c[(size_t)b * p.H + j], gi * gg);
        const float hn = go * tanhf(cn);
        p.c_out[(size_t)b * p.H + j] = cn;
        p.h_out[(size_t)b * p.H + j] = hn;
        if (p.series) p.series[b * p.series_stride_b + j] = hn;
        if (p.gates) {
            float *q = p.gates + (size_t)b * G;
            q[j] = gi; q[p.H + j] = gf; q[2 * p.H + j] = gg; q[3 * p.H + j] = go;
        }
    }
    if (p.series && p.series_cols > p.H) {
        const int np = p.series_cols - p.H;
        for (int i = tid; i < ns * np; i += kLstmThreads) {
            const int s = i / np, j = p.H + (i - s * np);
            p.series[(b0 + s) * p.series_stride_b + j] = p.pad_value;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The whole recurrence of LSTM_cell.forward as ONE launch (the reference's class counts, 33 / 38: I + H <= kSeriesK).
// One 256-thread workgroup per kSeriesSamples samples walks all T frames: thread r holds gate row r of [W_ih | W_hh] in
// REGISTERS for the whole launch, [x_t | h_{t-1}] of the samples sits in LDS (16-byte broadcast reads), the cell state
// stays in LDS between frames, x_{t+1} is in flight while frame t is computed, and h_t goes straight into v_series[t].
// Same fma chains, in the same order, as lstm_cell_step_kernel: the two paths agree bit for bit.  Two barriers per frame.
constexpr int kSeriesSamples = 4, kSeriesK = 80;

struct LstmSeriesParams {
    const float *x, *h0, *c0, *w_ih, *w_hh, *b_ih, *b_hh;
    int T, B, I, H;
    float *series;
    int64_t series_stride_t, series_stride_b;
    int series_cols;
    float pad_value;
    float *gates, *cells;                                    // optional [T][B][4H] activations, [T + 1][B][H] cell states (backward)
    float *h_out, *c_out;                                    // optional final state [B][H]
};

__global__ __launch_bounds__(kLstmThreads) void lstm_series_kernel(LstmSeriesParams p)
{
    extern __shared__ float4 series_smem[];
    const int K = p.I + p.H, G = 4 * p.H, K4 = (K + 3) >> 2;
    float *xh = reinterpret_cast<float *>(series_smem);      // [kSeriesSamples][4 * K4]: [x_t | h_{t-1} | zeros]
    float *pre = xh + kSeriesSamples * 4 * K4;               // [kSeriesSamples][G]
    float *cst = pre + kSeriesSamples * G;                   // [kSeriesSamples][H]
    const int tid = threadIdx.x, b0 = blockIdx.x * kSeriesSamples;
    const int ns = min(kSeriesSamples, p.B - b0);
    // gate row `tid` of [W_ih | W_hh] -> registers (zeros behind K: the staged vectors are padded alike)
    float wr[kSeriesK];
    float bias = 0.f;
    if (tid < G) {
#pragma unroll
        for (int k = 0; k < kSeriesK; ++k)
            wr[k] = k < p.I ? p.w_ih[(size_t)tid * p.I + k] : k < K ? p.w_hh[(size_t)tid * p.H + (k - p.I)] : 0.f;
        bias = p.b_ih[tid] + p.b_hh[tid];
    }
    for (int i = tid; i < kSeriesSamples * 4 * K4; i += kLstmThreads) {
        const int s = i / (4 * K4), k = i - s * 4 * K4;
        float v = 0.f;
        if (s < ns && k < p.I) v = p.x[(size_t)(b0 + s) * p.I + k];                       // x_0
        else if (s < ns && k < K) v = p.h0[(size_t)(b0 + s) * p.H + (k - p.I)];
        xh[i] = v;
    }
    for (int i = tid; i < kSeriesSamples * p.H; i += kLstmThreads) {
        const int s = i / p.H, j = i - s * p.H;
        const float c = s < ns ? p.c0[(size_t)(b0 + s) * p.H + j] : 0.f;
        cst[i] = c;
        if (p.cells && s < ns) p.cells[(size_t)(b0 + s) * p.H + j] = c;
    }
    // this thread's element of the next frame's inputs (kSeriesSamples * I <= 256 is checked on the host)
    const int xs = tid / p.I, xk = tid - xs * p.I;
    const bool xmine = xs < ns && tid < kSeriesSamples * p.I;
    // this thread's (sample, unit) of the cell update (kSeriesSamples * H <= 256)
    const int cs_ = tid / p.H, cj = tid - cs_ * p.H;
    const bool cmine = cs_ < ns && tid < kSeriesSamples * p.H;
    for (int t = 0; t < p.T; ++t) {
        __syncthreads();                                     // [x_t | h_{t-1}] is complete
        float xnext = 0.f;
        if (xmine && t + 1 < p.T) xnext = p.x[((size_t)(t + 1) * p.B + b0 + xs) * p.I + xk];
        if (tid < G) {
            float acc[kSeriesSamples];
#pragma unroll
            for (int s = 0; s < kSeriesSamples; ++s) acc[s] = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < kSeriesK / 4; ++k4) {
                if (k4 < K4) {                               // (uniform)
#pragma unroll
                    for (int s = 0; s < kSeriesSamples; ++s) {
                        const float4 v = *reinterpret_cast<const float4 *>(xh + (s * K4 + k4) * 4);
                        acc[s] = __builtin_fmaf(wr[4 * k4], v.x, acc[s]);
                        acc[s] = __builtin_fmaf(wr[4 * k4 + 1], v.y, acc[s]);
                        acc[s] = __builtin_fmaf(wr[4 * k4 + 2], v.z, acc[s]);
                        acc[s] = __builtin_fmaf(wr[4 * k4 + 3], v.w, acc[s]);
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < kSeriesSamples; ++s) pre[s * G + tid] = acc[s] + bias;
        }
        __syncthreads();                                     // the pre-activations are there; the staged vectors are free
        if (xmine) xh[xs * 4 * K4 + xk] = xnext;
        if (cmine) {
            const int b = b0 + cs_;
            const float *g4 = pre + cs_ * G;
            const float gi = sigmoid_f(g4[cj]), gf = sigmoid_f(g4[p.H + cj]), gg = tanhf(g4[2 * p.H + cj]), go = sigmoid_f(g4[3 * p.H + cj]);
            const float cn = __builtin_fmaf(gf, cst[cs_ * p.H + cj], gi * gg);
            const float hn = go * tanhf(cn);
            cst[cs_ * p.H + cj] = cn;
            xh[cs_ * 4 * K4 + p.I + cj] = hn;                // h_t: the next frame's recurrent input
            p.series[t * p.series_stride_t + b * p.series_stride_b + cj] = hn;
            if (p.gates) {
                float *q = p.gates + ((size_t)t * p.B + b) * G;
                q[cj] = gi; q[p.H + cj] = gf; q[2 * p.H + cj] = gg; q[3 * p.H + cj] = go;
            }
            if (p.cells) p.cells[((size_t)(t + 1) * p.B + b) * p.H + cj] = cn;
            if (t == p.T - 1) {
                if (p.h_out) p.h_out[(size_t)b * p.H + cj] = hn;
                if (p.c_out) p.c_out[(size_t)b * p.H + cj] = cn;
            }
        }
        if (p.series_cols > p.H) {
            const int np = p.series_cols - p.H;
            for (int i = tid; i < ns * np; i += kLstmThreads) {
                const int s = i / np, j = p.H + (i - s * np);
                p.series[t * p.series_stride_t + (b0 + s) * p.series_stride_b + j] = p.pad_value;
            }
        }
    }
}

// The backward recurrence of the same loop as one launch.  Only what is SEQUENTIAL stays in the kernel: thread (s, j) owns
// hidden unit j of sample s for all T frames -- dh and dc live in its registers -- turns (dh_t, dc_t) into the four
// pre-activation gradients of its unit (written out: [T][B][4H]), and takes dh_{t-1}(s, j) = sum_r W_hh[r][j] dpre(s, r) with
// column j of W_hh in registers and the pre-activation gradients of the sample's 4H gate rows read from LDS (16-byte
// broadcasts, double-buffered: one barrier per frame).  Everything that is not a recurrence -- dx_t = dpre_t W_ih, the three
// parameter gradients -- is a plain GEMM over all frames at once on the result (the caller: rocBLAS through torch).
constexpr int kSeriesG = 4 * 64;

struct LstmSeriesBwdParams {
    const float *d_series;                                   // upstream gradient of v_series, unit stride over classes
    int64_t ds_stride_t, ds_stride_b;
    const float *gates, *cells, *w_hh;                       // [T][B][4H] (i, f, g, o), [T + 1][B][H], [4H][H]
    int T, B, H;
    float *dpre, *dh0, *dc0;                                 // [T][B][4H], [B][H], [B][H]
};

__global__ __launch_bounds__(kLstmThreads) void lstm_series_bwd_kernel(LstmSeriesBwdParams p)
{
    extern __shared__ float4 bwd_smem[];
    const int G = 4 * p.H, G4 = p.H;                         // (G floats = H float4)
    float *dp = reinterpret_cast<float *>(bwd_smem);         // [2][kSeriesSamples][G]: the frame's pre-activation gradients
    const int tid = threadIdx.x, b0 = blockIdx.x * kSeriesSamples;
    const int ns = min(kSeriesSamples, p.B - b0);
    const int s = tid / p.H, j = tid - s * p.H;
    const bool mine = s < ns && tid < kSeriesSamples * p.H;
    const int b = b0 + (mine ? s : 0);
    float wc[kSeriesG];                                      // column j of W_hh
#pragma unroll
    for (int r = 0; r < kSeriesG; ++r) wc[r] = (mine && r < G) ? p.w_hh[(size_t)r * p.H + j] : 0.f;
    for (int i = tid; i < 2 * kSeriesSamples * G; i += kLstmThreads) dp[i] = 0.f;
    float dh = 0.f, dc = 0.f;                                // gradient arriving from frame t + 1
    // frame T - 1's operands (then always one frame ahead)
    auto fetch = [&](int t, float (&q)[4], float &ct, float &cp, float &ds) {
        const float *g = p.gates + ((size_t)t * p.B + b) * G;
        q[0] = g[j]; q[1] = g[p.H + j]; q[2] = g[2 * p.H + j]; q[3] = g[3 * p.H + j];
        ct = p.cells[((size_t)(t + 1) * p.B + b) * p.H + j];
        cp = p.cells[((size_t)t * p.B + b) * p.H + j];
        ds = p.d_series[t * p.ds_stride_t + b * p.ds_stride_b + j];
    };
    float q[4] = {0.f, 0.f, 0.f, 0.f}, ct = 0.f, cp = 0.f, ds = 0.f;
    if (mine) fetch(p.T - 1, q, ct, cp, ds);
    for (int t = p.T - 1; t >= 0; --t) {
        float *cur = dp + (size_t)(t & 1) * kSeriesSamples * G;
        float qn[4] = {0.f, 0.f, 0.f, 0.f}, ctn = 0.f, cpn = 0.f, dsn = 0.f;
        if (mine && t > 0) fetch(t - 1, qn, ctn, cpn, dsn);
        if (mine) {
            const float gi = q[0], gf = q[1], gg = q[2], go = q[3];
            const float dht = dh + ds;
            const float tc = tanhf(ct);
            const float dct = __builtin_fmaf(dht * go, 1.0f - tc * tc, dc);
            const float d_i = dct * gg * gi * (1.0f - gi);
            const float d_f = dct * cp * gf * (1.0f - gf);
            const float d_g = dct * gi * (1.0f - gg * gg);
            const float d_o = dht * tc * go * (1.0f - go);
            cur[s * G + j] = d_i; cur[s * G + p.H + j] = d_f; cur[s * G + 2 * p.H + j] = d_g; cur[s * G + 3 * p.H + j] = d_o;
            float *o = p.dpre + ((size_t)t * p.B + b) * G;
            o[j] = d_i; o[p.H + j] = d_f; o[2 * p.H + j] = d_g; o[3 * p.H + j] = d_o;
            dc = dct * gf;
        }
        __syncthreads();                                     // the frame's 4H pre-activation gradients of every sample are in LDS
        if (mine) {
            float acc = 0.f;
            const float4 *row = reinterpret_cast<const float4 *>(cur + s * G);
#pragma unroll
            for (int r4 = 0; r4 < kSeriesG / 4; ++r4) {
                if (r4 < G4) {                               // (uniform)
                    const float4 v = row[r4];
                    acc = __builtin_fmaf(wc[4 * r4], v.x, acc);
                    acc = __builtin_fmaf(wc[4 * r4 + 1], v.y, acc);
                    acc = __builtin_fmaf(wc[4 * r4 + 2], v.z, acc);
                    acc = __builtin_fmaf(wc[4 * r4 + 3], v.w, acc);
                }
            }
            dh = acc;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = qn[k];
        ct = ctn; cp = cpn; ds = dsn;
    }
    if (mine) {
        p.dh0[(size_t)b * p.H + j] = dh;
        p.dc0[(size_t)b * p.H + j] = dc;
    }
}

}  // namespace ctc

using namespace ctc;

// The T steps of ctc_amd_lstm_cell_step as one launch, for the reference's class counts (I + H <= 80, H <= 64, I <= 64).
// Other sizes: CTC_AMD_ERR_UNSUPPORTED_SHAPE (the caller steps frame by frame).
extern "C" int ctc_amd_lstm_series(const float *x, const float *h0, const float *c0,
                                   const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                                   int T, int B, int I, int H,
                                   float *series, int64_t series_stride_t, int64_t series_stride_b, int series_cols, float pad_value,
                                   float *gates_out, float *cells_out, float *h_out, float *c_out, void *stream)
{
    if (!x || !h0 || !c0 || !w_ih || !w_hh || !b_ih || !b_hh || !series) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || I < 1 || H < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (series_cols < H || series_stride_b < series_cols) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (I + H > kSeriesK || 4 * H > kLstmThreads || kSeriesSamples * I > kLstmThreads || kSeriesSamples * H > kLstmThreads)
        return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    LstmSeriesParams p;
    p.x = x; p.h0 = h0; p.c0 = c0; p.w_ih = w_ih; p.w_hh = w_hh; p.b_ih = b_ih; p.b_hh = b_hh;
    p.T = T; p.B = B; p.I = I; p.H = H;
    p.series = series; p.series_stride_t = series_stride_t; p.series_stride_b = series_stride_b;
    p.series_cols = series_cols; p.pad_value = pad_value;
    p.gates = gates_out; p.cells = cells_out; p.h_out = h_out; p.c_out = c_out;
    const size_t smem = (size_t)kSeriesSamples * (4 * (size_t)((I + H + 3) / 4) + 4 * (size_t)H + H) * sizeof(float);
    return launch<lstm_series_kernel>(dim3((B + kSeriesSamples - 1) / kSeriesSamples), dim3(kLstmThreads), smem,
                                      static_cast<hipStream_t>(stream), p);
}

extern "C" int ctc_amd_lstm_cell_step(const float *x, const float *h, const float *c,
                                      const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                                      int B, int I, int H,
                                      float *h_out, float *c_out, float *gates_out,
                                      float *series_row, int64_t series_stride_b, int series_cols, float pad_value,
                                      void *stream)
{
    if (!x || !h || !c || !w_ih || !w_hh || !b_ih || !b_hh || !h_out || !c_out) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (B < 1 || I < 1 || H < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (series_row && (series_cols < H || series_stride_b < series_cols)) return CTC_AMD_ERR_BAD_ARGUMENT;
    const size_t smem = (size_t)kLstmSamples * ((size_t)I + H + 4 * (size_t)H) * sizeof(float);
    if (smem > kMaxLds) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    LstmParams p;
    p.x = x; p.h = h; p.c = c; p.w_ih = w_ih; p.w_hh = w_hh; p.b_ih = b_ih; p.b_hh = b_hh;
    p.B = B; p.I = I; p.H = H;
    p.h_out = h_out; p.c_out = c_out; p.gates = gates_out;
    p.series = series_row; p.series_stride_b = series_stride_b; p.series_cols = series_cols; p.pad_value = pad_value;
    return launch<lstm_cell_step_kernel>(dim3((B + kLstmSamples - 1) / kLstmSamples), dim3(kLstmThreads), smem,
                                         static_cast<hipStream_t>(stream), p);
}


// The backward recurrence of ctc_amd_lstm_series (same sizes): from the upstream gradient of v_series and the state the
// forward launch saved to the pre-activation gradients of every frame and the gradients of (h0, c0).  The rest of the
// backward pass is three GEMMs on dpre (ctc_amd/producer.py).
extern "C" int ctc_amd_lstm_series_backward(const float *d_series, int64_t ds_stride_t, int64_t ds_stride_b,
                                            const float *gates, const float *cells, const float *w_hh,
                                            int T, int B, int H, float *dpre_out, float *dh0_out, float *dc0_out, void *stream)
{
    if (!d_series || !gates || !cells || !w_hh || !dpre_out || !dh0_out || !dc0_out) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || H < 1) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (4 * H > kSeriesG || kSeriesSamples * H > kLstmThreads) return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    LstmSeriesBwdParams p;
    p.d_series = d_series; p.ds_stride_t = ds_stride_t; p.ds_stride_b = ds_stride_b;
    p.gates = gates; p.cells = cells; p.w_hh = w_hh;
    p.T = T; p.B = B; p.H = H;
    p.dpre = dpre_out; p.dh0 = dh0_out; p.dc0 = dc0_out;
    const size_t smem = (size_t)2 * kSeriesSamples * 4 * H * sizeof(float);
    return launch<lstm_series_bwd_kernel>(dim3((B + kSeriesSamples - 1) / kSeriesSamples), dim3(kLstmThreads), smem,
                                          static_cast<hipStream_t>(stream), p);
}

// ---------------------------------------------------------------------------------------------------------------------
// The HEAD of the producer (LSTM.py:8-18, called per frame at :48): Linear(K -> C) + BatchNorm1d + ReLU +
// Dropout for ALL frames in one launch.  One workgroup per (frame, tile of 16 output columns); wave m owns batch rows
// 16 m .. 16 m + 15 (B <= 256 rows of one frame fit one workgroup, which is what BatchNorm's per-frame batch statistics
// need: a column's mean and variance are taken over the B rows of ONE frame, as the reference's per-frame calls do).
// The product runs on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32, an fmaf chain per output): a lane loads
// 16 bytes of its feature row and 16 bytes of its weight row per four MFMAs -- the k index is permuted the same way on
// both operands (lane (r, q) holds k = 16 kb + 4 q + i for MFMA i), which a sum over k does not see.
// Train mode: batch statistics (two passes: mean, then the centred second moment, like torch), saved per (frame, column)
// for the backward pass and for the running statistics, which the reference updates frame after frame (a closed form
// over the T frames, applied by the caller).  Eval mode: the running statistics.  Dropout is a mask tensor the caller
// hands in (already scaled by 1 / (1 - p)): the random stream stays torch's.
namespace ctc {

struct HeadParams {
    const float *feat;                                        // [T][B][K]
    int64_t fst, fsb;
    const float *w, *bias, *gamma, *beta;                    // Linear [C][K], [C]; BatchNorm weight / bias [C]
    const float *rmean, *rvar;                               // eval mode: running statistics; NULL: batch statistics
    const float *mask;                                       // [T][B][C] or NULL
    float eps;
    int T, B, K, C;
    float *out;                                              // [T][B] rows of C at (ost, osb)
    int64_t ost, osb;
    float *lin, *smean, *svar, *sinv;                        // optional: Linear output [T][B][C]; batch mean / biased variance / 1/sqrt(var + eps) [T][C]
};

typedef float head_f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void head_kernel(HeadParams p)
{
    __shared__ float red[16][16];                            // [wave][column of the tile]
    const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int NW = blockDim.x >> 6;
    // operands: A row = batch row 16 w + fr, B row = output column 16 n + fr (clamped: the padding is masked out below)
    const int arow = min(16 * w + fr, p.B - 1), bcol = min(16 * n + fr, p.C - 1);
    const float *ap = p.feat + (int64_t)t * p.fst + (int64_t)arow * p.fsb + 4 * fq;
    const float *bp = p.w + (int64_t)bcol * p.K + 4 * fq;
    head_f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int KB = p.K >> 4;                                 // K is a multiple of 16 (checked by the host)
    typedef head_f4 head_f4u;                                // (rows are 16-byte aligned: checked by the host)
    for (int kb = 0; kb < KB; kb += 2) {                     // two batches of four MFMAs in flight
        const head_f4 a0 = *reinterpret_cast<const head_f4u *>(ap + 16 * kb), b0 = *reinterpret_cast<const head_f4u *>(bp + 16 * kb);
        const bool two = kb + 1 < KB;
        const head_f4 a1 = two ? *reinterpret_cast<const head_f4u *>(ap + 16 * kb + 16) : head_f4{0.f, 0.f, 0.f, 0.f};
        const head_f4 b1 = two ? *reinterpret_cast<const head_f4u *>(bp + 16 * kb + 16) : head_f4{0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    // acc[j] = Linear output (without bias) of batch row 16 w + 4 fq + j, column 16 n + fr
    const int col = 16 * n + fr;
    const bool colok = col < p.C;
    const float bias = colok ? p.bias[col] : 0.f;
    float x[4];
    bool rowok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        rowok[j] = 16 * w + 4 * fq + j < p.B;
        x[j] = acc[j] + bias;
    }
    auto column_total = [&](float v) {                       // sum over the batch rows of the frame, every lane of a column gets it
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        __syncthreads();                                     // (the previous total has been read by everybody)
        if (fq == 0) red[w][fr] = v;
        __syncthreads();
        float s = 0.f;
        for (int i = 0; i < NW; ++i) s += red[i][fr];
        return s;
    };
    float mean, inv;
    if (p.rmean) {                                           // eval mode
        mean = colok ? p.rmean[col] : 0.f;
        inv = 1.0f / sqrtf((colok ? p.rvar[col] : 1.f) + p.eps);
    } else {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += rowok[j] ? x[j] : 0.f;
        mean = column_total(s) / (float)p.B;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) q += rowok[j] ? (x[j] - mean) * (x[j] - mean) : 0.f;
        const float var = column_total(q) / (float)p.B;      // biased, what the normalisation uses
        inv = 1.0f / sqrtf(var + p.eps);
        if (w == 0 && fq == 0 && colok) {
            if (p.smean) p.smean[(int64_t)t * p.C + col] = mean;
            if (p.svar) p.svar[(int64_t)t * p.C + col] = var;
            if (p.sinv) p.sinv[(int64_t)t * p.C + col] = inv;
        }
    }
    const float g = colok ? p.gamma[col] : 0.f, be = colok ? p.beta[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 16 * w + 4 * fq + j;
        if (!rowok[j] || !colok) continue;
        if (p.lin) p.lin[((int64_t)t * p.B + b) * p.C + col] = x[j];
        float y = (x[j] - mean) * inv * g + be;
        y = y > 0.f ? y : 0.f;
        if (p.mask) y *= p.mask[((int64_t)t * p.B + b) * p.C + col];
        p.out[(int64_t)t * p.ost + (int64_t)b * p.osb + col] = y;
    }
}

}  // namespace ctc

extern "C" int ctc_amd_head_forward(const float *feat, int64_t feat_stride_t, int64_t feat_stride_b,
                                    const float *weight, const float *bias, const float *bn_weight, const float *bn_bias,
                                    const float *running_mean, const float *running_var, float eps, const float *mask,
                                    int T, int B, int K, int C,
                                    float *out, int64_t out_stride_t, int64_t out_stride_b,
                                    float *linear_out, float *save_mean, float *save_var, float *save_invstd, void *stream)
{
    if (!feat || !weight || !bias || !bn_weight || !bn_bias || !out) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (T < 1 || B < 1 || K < 1 || C < 1 || (running_mean == nullptr) != (running_var == nullptr)) return CTC_AMD_ERR_BAD_ARGUMENT;
    if (out_stride_b < C) return CTC_AMD_ERR_BAD_ARGUMENT;
    // one workgroup holds the B rows of a frame (BatchNorm's statistics); 16-byte operand loads
    if (B > 256 || (K & 15) != 0 || (feat_stride_b & 3) != 0 || (feat_stride_t & 3) != 0 ||
        (reinterpret_cast<uintptr_t>(feat) & 15) != 0 || (reinterpret_cast<uintptr_t>(weight) & 15) != 0)
        return CTC_AMD_ERR_UNSUPPORTED_SHAPE;
    if (running_mean == nullptr && B < 2) return CTC_AMD_ERR_BAD_ARGUMENT;       // (torch raises too: one value per channel)
    ctc::HeadParams p;
    p.feat = feat; p.fst = feat_stride_t; p.fsb = feat_stride_b;
    p.w = weight; p.bias = bias; p.gamma = bn_weight; p.beta = bn_bias;
    p.rmean = running_mean; p.rvar = running_var; p.mask = mask; p.eps = eps;
    p.T = T; p.B = B; p.K = K; p.C = C;
    p.out = out; p.ost = out_stride_t; p.osb = out_stride_b;
    p.lin = linear_out; p.smean = save_mean; p.svar = save_var; p.sinv = save_invstd;
    const int NW = (B + 15) / 16;
    hipLaunchKernelGGL(ctc::head_kernel, dim3(T, (C + 15) / 16), dim3(64 * NW), 0, static_cast<hipStream_t>(stream), p);
    return (int)hipGetLastError();
}
