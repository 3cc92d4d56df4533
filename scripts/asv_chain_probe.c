/* Developer probe (CPU): how many (cell, restricted cell) pairs of adjust_shift_variance's three log-sum chains
 * (src/adjust_shift_variance.cpp:74-157) can change a chain at all.  An addend so far below the running sum that
 * fl(acc + log1p(exp(lw - acc))) == acc is an exact no-op; leaving such addends out gives the same bits.  The probe runs
 * the full chains and the chains over the kept addends side by side and counts.
 *   gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC scripts/asv_chain_probe.c -o /tmp/libasvprobe.so -lm
 * TEST INFRASTRUCTURE: includes the oracle's source for its static helpers. */
#include "../oracle/mnn_oracle.c"

#define LN2 0.6931471805599453

/* may an addend lw be left out when the running sum is known to lie in [L, U]?  (see legacy.hip: asv_skippable) */
static int skippable(double lw, double L, double U) {
    if (lw < L - 701.0) return 1; /* exp(lw - acc) is 0 in the portable exp (cut at -700): log1p(0) = 0 */
    if (!((L > 0.0 && U > 0.0) || (L < 0.0 && U < 0.0))) return 0;
    const double amin = fabs(L) < fabs(U) ? fabs(L) : fabs(U);
    int e;
    frexp(amin, &e); /* amin = f * 2^e, f in [0.5, 1): floor(log2 amin) = e - 1 */
    return lw < L + (double)(e - 1 - 54) * LN2 - 0.25;
}

typedef struct {
    double proj, logw;
    int add;
} el_t;

int probe_cells(const double* data1, const double* data2, int32_t g, int32_t n1, int32_t n2, const double* vect,
                double sigma2, const int32_t* r1, int32_t nr1, const int32_t* r2, int32_t nr2, const int32_t* cells,
                int32_t ncells, double* out_full, double* out_sub, int64_t* K, double* ulps) {
    (void)n1;
#pragma omp parallel
    {
        double* work = (double*)malloc(((size_t)g + 1) * sizeof(double));
        double* grad = (double*)malloc(((size_t)g + 1) * sizeof(double));
        pw_t* d1 = (pw_t*)malloc(((size_t)nr1 + 1) * sizeof(pw_t));
        el_t* own = (el_t*)malloc(((size_t)nr2 + 1) * sizeof(el_t));
#pragma omp for schedule(dynamic, 1)
        for (int32_t ci = 0; ci < ncells; ++ci) {
            const int32_t cell = cells[ci];
            const double* cur = data2 + (size_t)cell * g;
            double l2 = 0.0;
            for (int32_t x = 0; x < g; ++x) {
                grad[x] = vect[(size_t)x * n2 + cell];
                l2 += grad[x] * grad[x];
            }
            l2 = sqrt(l2);
            if (l2 != 0.0)
                for (int32_t x = 0; x < g; ++x) grad[x] /= l2;
            const double curproj = dot(grad, cur, g);
            for (int32_t s = 0; s < nr2; ++s) {
                const int32_t same = r2[s];
                own[s].add = 1;
                own[s].logw = 0.0;
                own[s].proj = curproj;
                if (same != cell) {
                    const double* sc = data2 + (size_t)same * g;
                    own[s].proj = dot(grad, sc, g);
                    own[s].logw = -dist2_to_line(cur, grad, sc, work, g) / sigma2;
                    own[s].add = !(own[s].proj > curproj);
                }
            }
            /* full chains */
            double prob2 = 0.0, tot2 = 0.0;
            int fp = 1;
            for (int32_t s = 0; s < nr2; ++s) {
                if (own[s].add) {
                    prob2 = fp ? own[s].logw : logspace_add(prob2, own[s].logw);
                    fp = 0;
                }
                tot2 = s == 0 ? own[s].logw : logspace_add(tot2, own[s].logw);
            }
            /* kept chains: bounds from the running maximum (lower) and the count (upper) */
            double sp = 0.0, st = 0.0;
            int sfp = 1, sft = 1;
            int64_t k2 = 0;
            {
                double Mp = -INFINITY, Mt = -INFINITY; /* running maxima of the addends of each chain */
                double Sp = 0.0, St = 0.0;             /* lower bounds on exp(acc): the largest addend so far, plus self */
                int64_t np_ = 0, nt_ = 0;
                int self_seen = 0;
                for (int32_t s = 0; s < nr2; ++s) {
                    const double lw = own[s].logw;
                    int keep = 0;
                    /* tot2 */
                    {
                        double L = Mt, U = Mt + log((double)nt_ + 1.0) + 0.01;
                        if (self_seen && St > 0.0) { /* the cell itself (log-weight 0) has gone by: acc = log1p(sum of the rest) */
                            L = log1p(St) * (1.0 - 1e-9);
                            U = log1p(St * ((double)nt_ + 1.0)) * (1.0 + 1e-9) + 1e-300;
                        }
                        if (nt_ == 0 || !skippable(lw, L, U)) keep |= 1;
                    }
                    if (own[s].add) {
                        double L = Mp, U = Mp + log((double)np_ + 1.0) + 0.01;
                        if (self_seen && Sp > 0.0) {
                            L = log1p(Sp) * (1.0 - 1e-9);
                            U = log1p(Sp * ((double)np_ + 1.0)) * (1.0 + 1e-9) + 1e-300;
                        }
                        if (np_ == 0 || !skippable(lw, L, U)) keep |= 2;
                    }
                    /* bounds advance with EVERY addend (kept or not: a skipped one is below them anyway) */
                    if (r2[s] != cell) {
                        const double w = exp(lw);
                        if (w > St) St = w;
                        if (own[s].add && w > Sp) Sp = w;
                    } else {
                        self_seen = 1;
                    }
                    if (lw > Mt) Mt = lw;
                    ++nt_;
                    if (own[s].add) {
                        if (lw > Mp) Mp = lw;
                        ++np_;
                    }
                    if (keep) {
                        ++k2;
                        if (own[s].add) {
                            sp = sfp ? lw : logspace_add(sp, lw);
                            sfp = 0;
                        }
                        st = sft ? lw : logspace_add(st, lw);
                        sft = 0;
                    }
                }
            }
            prob2 -= tot2;
            sp -= st;

            double tot1 = 0.0, stot1 = 0.0, G = -INFINITY;
            int64_t k1r = 0, k1s = 0, cg = 0;
            {
                double M = -INFINITY;
                int sf = 1;
                for (int32_t o = 0; o < nr1; ++o) {
                    const double* oc = data1 + (size_t)r1[o] * g;
                    d1[o].proj = dot(grad, oc, g);
                    d1[o].logw = -dist2_to_line(cur, grad, oc, work, g) / sigma2;
                    tot1 = o == 0 ? d1[o].logw : logspace_add(tot1, d1[o].logw);
                    const double lw = d1[o].logw;
                    if (o == 0 || !skippable(lw, M, M + log((double)o + 1.0) + 0.01)) {
                        ++k1r;
                        stot1 = sf ? lw : logspace_add(stot1, lw);
                        sf = 0;
                    }
                    if (lw > M) M = lw;
                }
                G = M;
            }
            for (int32_t o = 0; o < nr1; ++o) cg += d1[o].logw >= G - 38.0;
            qsort(d1, (size_t)nr1, sizeof(pw_t), pw_cmp);
            double rq = NAN, srq = NAN;
            if (nr1 > 0) {
                const double target = prob2 + tot1, starget = sp + stot1;
                double cum = 0.0;
                rq = d1[nr1 - 1].proj;
                for (int32_t o = 0; o < nr1; ++o) {
                    cum = o == 0 ? d1[o].logw : logspace_add(cum, d1[o].logw);
                    if (cum >= target) {
                        rq = d1[o].proj;
                        break;
                    }
                }
                double scum = 0.0, M = -INFINITY;
                int sf = 1;
                srq = d1[nr1 - 1].proj;
                for (int32_t o = 0; o < nr1; ++o) {
                    const double lw = d1[o].logw;
                    if (o == 0 || !skippable(lw, M, M + log((double)o + 1.0) + 0.01)) {
                        ++k1s;
                        scum = sf ? lw : logspace_add(scum, lw);
                        sf = 0;
                        if (scum >= starget) {
                            srq = d1[o].proj;
                            break;
                        }
                    }
                    if (lw > M) M = lw;
                }
                ulps[ci] = (tot1 - target) / (nextafter(fabs(tot1), INFINITY) - fabs(tot1));
            }
            out_full[ci] = (rq - curproj) / l2;
            out_sub[ci] = (srq - curproj) / l2;
            K[4 * ci + 0] = k2;
            K[4 * ci + 1] = k1r;
            K[4 * ci + 2] = k1s;
            K[4 * ci + 3] = cg;
        }
        free(work);
        free(grad);
        free(d1);
        free(own);
    }
    return 0;
}
