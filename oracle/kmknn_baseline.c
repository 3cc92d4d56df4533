/*
 * kmknn_baseline.c -- CPU baseline "A" for bench.py: a single-threaded, pruned, EXACT k-nearest-neighbour search
 * in the manner of BiocNeighbors' KmknnParam(), which is what fastMNN() runs by default
 * (BNPARAM=KmknnParam(), BPPARAM=SerialParam(): R/fastMNN.R:287).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY: never linked into the product.  BiocNeighbors is a third-party dependency whose
 * source is not under /root/reference; this is a restatement of the published algorithm it implements (KMKNN:
 * Wang, "A fast exact k-nearest neighbors algorithm for high dimensional search using k-means clustering and
 * triangle inequality", IJCNN 2011): k-means with ceil(sqrt(N)) centres over the reference points; members of a
 * cluster sorted by their distance to the centre; a query visits the clusters in order of centre distance and inside
 * a cluster only the members p with |d(q, c) - d(p, c)| below the current k-th distance (triangle inequality).
 * Results are exact; tests/test_oracle_kat.py checks them against the brute-force oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t d, n, nc;
    double* centres;   /* [nc][d] */
    int32_t* start;    /* [nc + 1] */
    int32_t* member;   /* [n] original ids, cluster-major, ascending distance to the centre */
    double* mdist;     /* [n] distance of member to its centre */
    double* pts;       /* [n][d] points in member order (contiguous scans) */
} kmknn_t;

static double dist2(const double* a, const double* b, int d) {
    double s = 0.0;
    for (int c = 0; c < d; ++c) {
        const double t = a[c] - b[c];
        s += t * t;
    }
    return s;
}

typedef struct {
    double d;
    int32_t i;
} di_t;
static int di_cmp(const void* a, const void* b) {
    const di_t* x = (const di_t*)a;
    const di_t* y = (const di_t*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}

void kmknn_free(kmknn_t* t) {
    if (!t) return;
    free(t->centres);
    free(t->start);
    free(t->member);
    free(t->mdist);
    free(t->pts);
    free(t);
}

/* Lloyd iterations from evenly strided starting centres (deterministic). */
kmknn_t* kmknn_build(const double* X, int32_t n, int32_t d, int32_t iters) {
    kmknn_t* t = (kmknn_t*)calloc(1, sizeof(kmknn_t));
    if (!t) return NULL;
    int32_t nc = (int32_t)ceil(sqrt((double)n));
    if (nc < 1) nc = 1;
    t->d = d;
    t->n = n;
    t->nc = nc;
    t->centres = (double*)malloc((size_t)nc * d * sizeof(double));
    t->start = (int32_t*)calloc((size_t)nc + 1, sizeof(int32_t));
    t->member = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    t->mdist = (double*)malloc((size_t)n * sizeof(double));
    t->pts = (double*)malloc((size_t)n * d * sizeof(double));
    int32_t* assign = (int32_t*)malloc((size_t)n * sizeof(int32_t));
    int32_t* count = (int32_t*)malloc((size_t)nc * sizeof(int32_t));
    di_t* tmp = (di_t*)malloc((size_t)n * sizeof(di_t));
    if (!t->centres || !t->start || !t->member || !t->mdist || !t->pts || !assign || !count || !tmp) {
        free(assign);
        free(count);
        free(tmp);
        kmknn_free(t);
        return NULL;
    }
    for (int32_t c = 0; c < nc; ++c) memcpy(t->centres + (size_t)c * d, X + (size_t)((int64_t)c * n / nc) * d, d * sizeof(double));
    for (int32_t it = 0; it <= iters; ++it) {
        for (int32_t i = 0; i < n; ++i) {
            double best = INFINITY;
            int32_t bc = 0;
            for (int32_t c = 0; c < nc; ++c) {
                const double s = dist2(X + (size_t)i * d, t->centres + (size_t)c * d, d);
                if (s < best) {
                    best = s;
                    bc = c;
                }
            }
            assign[i] = bc;
        }
        if (it == iters) break;
        memset(count, 0, (size_t)nc * sizeof(int32_t));
        double* sum = (double*)calloc((size_t)nc * d, sizeof(double));
        if (!sum) break;
        for (int32_t i = 0; i < n; ++i) {
            ++count[assign[i]];
            for (int32_t c = 0; c < d; ++c) sum[(size_t)assign[i] * d + c] += X[(size_t)i * d + c];
        }
        for (int32_t c = 0; c < nc; ++c)
            if (count[c] > 0)
                for (int32_t x = 0; x < d; ++x) t->centres[(size_t)c * d + x] = sum[(size_t)c * d + x] / count[c];
        free(sum);
    }
    /* cluster-major member lists sorted by distance to the centre */
    memset(count, 0, (size_t)nc * sizeof(int32_t));
    for (int32_t i = 0; i < n; ++i) ++count[assign[i]];
    for (int32_t c = 0; c < nc; ++c) t->start[c + 1] = t->start[c] + count[c];
    memset(count, 0, (size_t)nc * sizeof(int32_t));
    for (int32_t i = 0; i < n; ++i) {
        const int32_t c = assign[i];
        const int32_t pos = t->start[c] + count[c]++;
        tmp[pos].i = i;
        tmp[pos].d = sqrt(dist2(X + (size_t)i * d, t->centres + (size_t)c * d, d));
    }
    for (int32_t c = 0; c < nc; ++c) qsort(tmp + t->start[c], (size_t)(t->start[c + 1] - t->start[c]), sizeof(di_t), di_cmp);
    for (int32_t p = 0; p < n; ++p) {
        t->member[p] = tmp[p].i;
        t->mdist[p] = tmp[p].d;
        memcpy(t->pts + (size_t)p * d, X + (size_t)tmp[p].i * d, d * sizeof(double));
    }
    free(assign);
    free(count);
    free(tmp);
    return t;
}

/* k nearest of each query (ties by lowest index); idx 0-based [nq][k], dist Euclidean [nq][k] ascending.
 * *evals receives the number of point-to-query distances actually computed (the pruning statistic). */
int kmknn_query(const kmknn_t* t, const double* Q, int32_t nq, int32_t k, int32_t* idx, double* dist, int64_t* evals) {
    const int32_t d = t->d, nc = t->nc;
    if (k > t->n) return -1;
    di_t* cd = (di_t*)malloc((size_t)nc * sizeof(di_t));
    di_t* heap = (di_t*)malloc((size_t)k * sizeof(di_t)); /* kept unsorted with the worst tracked: k is small */
    if (!cd || !heap) {
        free(cd);
        free(heap);
        return -6;
    }
    int64_t ev = 0;
    for (int32_t q = 0; q < nq; ++q) {
        const double* qv = Q + (size_t)q * d;
        for (int32_t c = 0; c < nc; ++c) {
            cd[c].d = sqrt(dist2(qv, t->centres + (size_t)c * d, d));
            cd[c].i = c;
        }
        qsort(cd, (size_t)nc, sizeof(di_t), di_cmp);
        int32_t have = 0, worst = 0;
        double kth = INFINITY; /* current k-th squared distance */
        for (int32_t ci = 0; ci < nc; ++ci) {
            const int32_t c = cd[ci].i;
            const double dc = cd[ci].d;
            const int32_t s0 = t->start[c], s1 = t->start[c + 1];
            if (s1 == s0) continue;
            const double kd = sqrt(kth);
            /* whole cluster out of reach: its farthest member is still too close to the centre */
            if (have == k && dc - t->mdist[s1 - 1] > kd) continue;
            for (int32_t p = s1 - 1; p >= s0; --p) { /* from the rim inwards */
                const double lower = fabs(dc - t->mdist[p]);
                if (have == k && lower > sqrt(kth)) {
                    if (t->mdist[p] < dc) break; /* everything further in is even farther from the query */
                    continue;
                }
                const double s = dist2(qv, t->pts + (size_t)p * d, d);
                ++ev;
                const int32_t id = t->member[p];
                if (have < k) {
                    heap[have].d = s;
                    heap[have].i = id;
                    ++have;
                    if (have == k) {
                        worst = 0;
                        for (int32_t j = 1; j < k; ++j)
                            if (heap[j].d > heap[worst].d || (heap[j].d == heap[worst].d && heap[j].i > heap[worst].i)) worst = j;
                        kth = heap[worst].d;
                    }
                } else if (s < heap[worst].d || (s == heap[worst].d && id < heap[worst].i)) {
                    heap[worst].d = s;
                    heap[worst].i = id;
                    worst = 0;
                    for (int32_t j = 1; j < k; ++j)
                        if (heap[j].d > heap[worst].d || (heap[j].d == heap[worst].d && heap[j].i > heap[worst].i)) worst = j;
                    kth = heap[worst].d;
                }
            }
        }
        qsort(heap, (size_t)k, sizeof(di_t), di_cmp);
        for (int32_t j = 0; j < k; ++j) {
            idx[(size_t)q * k + j] = heap[j].i;
            dist[(size_t)q * k + j] = sqrt(heap[j].d);
        }
    }
    if (evals) *evals = ev;
    free(cd);
    free(heap);
    return 0;
}
