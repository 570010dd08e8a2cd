/* CPU oracle, C restatement -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Included twice by ctc_oracle.c (REAL=float, SUF=_f32 and REAL=double, SUF=_f64).
 * Same arithmetic as oracle/ctc_numpy.py (which is pinned against the golden
 * vectors captured from the reference); tests/test_oracle.py checks this file
 * against the same fixtures.  Batch loop is OpenMP-parallel (samples are
 * independent: computes_transition has no cross-b term, NoBlankCTC.py:71-87).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

static inline REAL FN(lse2)(REAL a, REAL b)
{   /* _logsumexp over a 2-stack, NoBlankCTC.py:16-19 */
    REAL m = a > b ? a : b;
    return m + (REAL)LOGF((REAL)EXPF(a - m) + (REAL)EXPF(b - m));
}

/* alpha / beta' / gamma on emissions e[T*S] of one sample (computes_transition,
 * NoBlankCTC.py:71-87; readout NoBlankCTC.py:58-68,139).  Returns nll; writes
 * gamma[T*S] (0 outside the live region) when gam != NULL. */
static REAL FN(lattice)(const REAL *e, int T, int S, long Tb, long L, REAL *al, REAL *be, REAL *gam)
{
    const REAL neg = (REAL)-10000000000000.0;      /* zero_padding, NoBlankCTC.py:25 */
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < S; ++l) {
            REAL stay = t ? al[(t - 1) * S + l] : (l == 0 ? (REAL)0 : neg);
            REAL adv = (t > 0 && l > 0) ? al[(t - 1) * S + l - 1] : neg;
            REAL p = FN(lse2)(stay, adv);
            if (l >= L) p = neg;
            al[t * S + l] = p + e[t * S + l];
        }
    long tb = ((Tb - 1) % T + T) % T, lb = ((L - 1) % S + S) % S;
    REAL nll = -al[tb * S + lb];
    if (!gam) return nll;
    for (long t = tb; t >= 0; --t)
        for (int l = 0; l < S; ++l) {
            REAL stay = (t < tb) ? be[(t + 1) * S + l] : (l == lb ? (REAL)0 : neg);
            REAL adv = (t < tb && l + 1 < S) ? be[(t + 1) * S + l + 1] : neg;
            REAL p = FN(lse2)(stay, adv);
            if (l >= L) p = neg;
            be[t * S + l] = p + e[t * S + l];
        }
    int feasible = nll < (REAL)1e12;
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < S; ++l) {
            REAL g = 0;
            if (feasible && t <= tb && t < Tb && l < L)
                g = (REAL)EXPF(al[t * S + l] + be[t * S + l] - e[t * S + l] + nll);
            gam[t * S + l] = g;
        }
    return nll;
}

/* NoBlankCTC.forward, NoBlankCTC.py:129-141 (+ closed-form input gradient).
 * x[T,B,C] logits, lab[B,S] (int64; values beyond L_b ignored, -1 -> class C-1),
 * nll[B], grad[T,B,C] or NULL, scale = gradient scale (1/B for the plain mean). */
void FN(oracle_noblank)(const REAL *x, const long *lab, const long *in_len, const long *tgt_len,
                        int T, int B, int C, int S, REAL scale, REAL *nll, REAL *grad, int threads)
{
#pragma omp parallel num_threads(threads)
    {
        REAL *lse = malloc(sizeof(REAL) * T), *mx = malloc(sizeof(REAL) * T);
        REAL *e = malloc(sizeof(REAL) * T * S), *al = malloc(sizeof(REAL) * T * S);
        REAL *be = malloc(sizeof(REAL) * T * S), *gam = malloc(sizeof(REAL) * T * S);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            for (int t = 0; t < T; ++t) {           /* LogSoftmax(dim=2), NoBlankCTC.py:136 */
                const REAL *r = x + ((long)t * B + b) * C;
                REAL m = r[0], s = 0;
                for (int c = 1; c < C; ++c) m = r[c] > m ? r[c] : m;
                for (int c = 0; c < C; ++c) s += (REAL)EXPF(r[c] - m);
                mx[t] = m; lse[t] = (REAL)LOGF(s);
                for (int l = 0; l < S; ++l) {       /* emission gather, NoBlankCTC.py:96-102 */
                    long k = lab[(long)b * S + l];
                    k = ((k % C) + C) % C;
                    e[t * S + l] = (r[k] - m) - lse[t];
                }
            }
            nll[b] = FN(lattice)(e, T, S, in_len[b], tgt_len[b], al, be, grad ? gam : NULL);
            if (!grad) continue;
            for (int t = 0; t < T; ++t) {
                const REAL *r = x + ((long)t * B + b) * C;
                REAL *g = grad + ((long)t * B + b) * C;
                if (t >= in_len[b] || !(nll[b] < (REAL)1e12)) { for (int c = 0; c < C; ++c) g[c] = 0; continue; }
                for (int c = 0; c < C; ++c) g[c] = (REAL)EXPF((r[c] - mx[t]) - lse[t]);
                for (int l = 0; l < S && l < tgt_len[b]; ++l) {
                    long k = lab[(long)b * S + l];
                    k = ((k % C) + C) % C;
                    g[k] -= gam[t * S + l];
                }
                for (int c = 0; c < C; ++c) g[c] *= scale;
            }
        }
        free(lse); free(mx); free(e); free(al); free(be); free(gam);
    }
}

/* NoBlankBinaryCTC.forward, NoBlankBinaryCTC.py:139-151: sigmoid (:146), emission
 * = -BCELoss(p[t,b,:], y[b,l,:]) (:112,:88), logs clamped at -100 like torch.
 * y[B,S,C]; scale = 1/B (the 1/C factor is applied here). */
void FN(oracle_binary)(const REAL *x, const REAL *y, const long *in_len, const long *tgt_len,
                       int T, int B, int C, int S, REAL scale, REAL *nll, REAL *grad, int threads)
{
#pragma omp parallel num_threads(threads)
    {
        REAL *e = malloc(sizeof(REAL) * T * S), *al = malloc(sizeof(REAL) * T * S);
        REAL *be = malloc(sizeof(REAL) * T * S), *gam = malloc(sizeof(REAL) * T * S);
        REAL *lp = malloc(sizeof(REAL) * C), *lq = malloc(sizeof(REAL) * C);
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b) {
            for (int t = 0; t < T; ++t) {
                const REAL *r = x + ((long)t * B + b) * C;
                for (int c = 0; c < C; ++c) {
                    REAL p = (REAL)1 / ((REAL)1 + (REAL)EXPF(-r[c]));
                    REAL a = (REAL)LOGF(p), q = (REAL)LOGF((REAL)1 - p);
                    lp[c] = a < (REAL)-100 ? (REAL)-100 : a;
                    lq[c] = q < (REAL)-100 ? (REAL)-100 : q;
                }
                for (int l = 0; l < S; ++l) {
                    const REAL *yy = y + ((long)b * S + l) * C;
                    REAL acc = 0;
                    for (int c = 0; c < C; ++c) acc += yy[c] * lp[c] + ((REAL)1 - yy[c]) * lq[c];
                    e[t * S + l] = acc / (REAL)C;
                }
            }
            nll[b] = FN(lattice)(e, T, S, in_len[b], tgt_len[b], al, be, grad ? gam : NULL);
            if (!grad) continue;
            for (int t = 0; t < T; ++t) {
                const REAL *r = x + ((long)t * B + b) * C;
                REAL *g = grad + ((long)t * B + b) * C;
                if (t >= in_len[b] || !(nll[b] < (REAL)1e12)) { for (int c = 0; c < C; ++c) g[c] = 0; continue; }
                REAL tot = 0;
                for (int l = 0; l < S; ++l) tot += gam[t * S + l];
                for (int c = 0; c < C; ++c) {
                    REAL p = (REAL)1 / ((REAL)1 + (REAL)EXPF(-r[c]));
                    REAL occ = 0;
                    for (int l = 0; l < S && l < tgt_len[b]; ++l)
                        occ += gam[t * S + l] * y[((long)b * S + l) * C + c];
                    REAL pq = p * ((REAL)1 - p);
                    REAL ratio = pq / (pq > (REAL)1e-12 ? pq : (REAL)1e-12);
                    g[c] = scale / (REAL)C * (p * tot - occ) * ratio;
                }
            }
        }
        free(e); free(al); free(be); free(gam); free(lp); free(lq);
    }
}

static inline REAL FN(lse3)(REAL a, REAL b, REAL c)
{
    REAL m = a > b ? a : b;
    m = m > c ? m : c;
    if (m == -(REAL)INFINITY) return m;
    return m + (REAL)LOGF((REAL)EXPF(a - m) + (REAL)EXPF(b - m) + (REAL)EXPF(c - m));
}

/* torch.nn.CTCLoss(blank, reduction='mean', zero_infinity=False) semantics
 * (models/layers/AsyncTFCriterion.py:198,319-321; arithmetic = aten::_ctc_loss,
 * third-party).  lp[T,B,C] log-probs, tgt[B,S] padded, nll[B] (un-normalised),
 * grad[T,B,C] or NULL = (exp(lp) - occupancy) * gscale[b]. */
void FN(oracle_blank)(const REAL *lp, const long *tgt, const long *in_len, const long *tgt_len,
                      int T, int B, int C, int S, int blank, const REAL *gscale, REAL *nll,
                      REAL *grad, int threads)
{
    const REAL ninf = -(REAL)INFINITY;
    int NS = 2 * S + 1;
#pragma omp parallel num_threads(threads)
    {
        REAL *al = malloc(sizeof(REAL) * (size_t)T * NS), *be = malloc(sizeof(REAL) * (size_t)T * NS);
        REAL *occ = malloc(sizeof(REAL) * C);
        long *ext = malloc(sizeof(long) * NS);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) {
            long Tb = in_len[b], L = tgt_len[b];
            int n = (int)(2 * L + 1);
            for (int s = 0; s < n; ++s) ext[s] = (s & 1) ? tgt[(long)b * S + s / 2] : blank;
#define LP(t, s) lp[((long)(t) * B + b) * C + ext[s]]
            for (int s = 0; s < n; ++s) al[s] = ninf;
            if (Tb > 0) { al[0] = LP(0, 0); if (n > 1) al[1] = LP(0, 1); }
            for (long t = 1; t < Tb; ++t)
                for (int s = 0; s < n; ++s) {
                    REAL a0 = al[(t - 1) * NS + s];
                    REAL a1 = s > 0 ? al[(t - 1) * NS + s - 1] : ninf;
                    REAL a2 = (s > 1 && ext[s] != blank && ext[s] != ext[s - 2]) ? al[(t - 1) * NS + s - 2] : ninf;
                    al[t * NS + s] = FN(lse3)(a0, a1, a2) + LP(t, s);
                }
            REAL ll;
            if (Tb == 0) ll = (L == 0) ? 0 : ninf;
            else ll = FN(lse3)(al[(Tb - 1) * NS + n - 1], n > 1 ? al[(Tb - 1) * NS + n - 2] : ninf, ninf);
            nll[b] = -ll;
            if (!grad) continue;
            for (long t = 0; t < T; ++t) {
                REAL *g = grad + ((long)t * B + b) * C;
                for (int c = 0; c < C; ++c) g[c] = 0;
            }
            if (Tb == 0) continue;
            for (int s = 0; s < n; ++s) be[(Tb - 1) * NS + s] = ninf;
            be[(Tb - 1) * NS + n - 1] = LP(Tb - 1, n - 1);
            if (n > 1) be[(Tb - 1) * NS + n - 2] = LP(Tb - 1, n - 2);
            for (long t = Tb - 2; t >= 0; --t)
                for (int s = 0; s < n; ++s) {
                    REAL b0 = be[(t + 1) * NS + s];
                    REAL b1 = s + 1 < n ? be[(t + 1) * NS + s + 1] : ninf;
                    REAL b2 = (s + 2 < n && ext[s + 2] != blank && ext[s + 2] != ext[s]) ? be[(t + 1) * NS + s + 2] : ninf;
                    be[t * NS + s] = FN(lse3)(b0, b1, b2) + LP(t, s);
                }
            for (long t = 0; t < Tb; ++t) {
                REAL *g = grad + ((long)t * B + b) * C;
                const REAL *row = lp + ((long)t * B + b) * C;
                for (int c = 0; c < C; ++c) occ[c] = ninf;
                for (int s = 0; s < n; ++s) {
                    REAL v = al[t * NS + s] + be[t * NS + s];
                    REAL o = occ[ext[s]];
                    occ[ext[s]] = FN(lse3)(o, v, ninf);
                }
                for (int c = 0; c < C; ++c)
                    g[c] = ((REAL)EXPF(row[c]) - (REAL)EXPF(occ[c] + nll[b] - row[c])) * gscale[b];
            }
#undef LP
        }
        free(al); free(be); free(occ); free(ext);
    }
}

#undef FN
#undef CAT
#undef CAT_
