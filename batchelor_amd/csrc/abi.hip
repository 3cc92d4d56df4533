// extern "C" boundary of libbatchelor_mi355x.so (declared in include/batchelor_mi355x.h).
// Nothing throws across it: exceptions become return codes + a thread-local message.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstddef>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "bmx_common.hpp"
#include "bmx_ops.hpp"
#include "engine.hpp"
#include "host_xfer.hpp"
#include "rccl_dyn.hpp"

struct bmx_engine {
    std::unique_ptr<bmx::Engine> impl;
};

namespace {

thread_local std::string g_last_error;
thread_local int64_t g_last_fallbacks = 0;
thread_local int g_force_exact = 0;
thread_local std::unique_ptr<bmx::Engine> g_prim;  // scratch engine behind the single-primitive entry points

int guarded(const std::function<void()>& fn) {
    try {
        fn();
        return BMX_OK;
    } catch (const bmx::Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "out of host memory";
        return BMX_ERR_HIP;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return BMX_ERR_ARG;
    }
}

bmx::Engine& prim(int d) {
    if (!g_prim) {
        int dev = 0;
        BMX_HIP(hipGetDevice(&dev));
        g_prim = std::make_unique<bmx::Engine>(dev);
    }
    g_prim->d_ = d;
    g_prim->knn_ws_.force_exact = g_force_exact;
    return *g_prim;
}

template <class T>
T* upload(bmx::DevBuf<T>& buf, const T* host, size_t n, hipStream_t s) {
    T* p = buf.reserve(std::max<size_t>(n, 1));
    // (the caller's memory is pageable: anything sizeable goes through the pinned staging ring at link speed -- a plain
    // hipMemcpyAsync of it is staged by the runtime at 4-5 GB/s, 8 of the 12 ms the smooth_gaussian_kernel call spent off the GPU)
    if (n * sizeof(T) >= ((size_t)1 << 20)) bmx::upload_pageable(p, host, n * sizeof(T), s);
    else if (n) BMX_HIP(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    return p;
}

// neighbour lists [nq][k] (0-based, row-major) -> R's layout [k][nq] (1-based, column-major), on the device
__global__ void idx_to_r_layout(const int32_t* __restrict__ in, int nq, int k, int32_t* __restrict__ out) {
    __shared__ int32_t tile[32][33];
    const int q0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int qq = threadIdx.y; qq < 32; qq += 8) {
        const int q = q0 + qq, j = j0 + threadIdx.x;
        if (q < nq && j < k) tile[qq][threadIdx.x] = in[(int64_t)q * k + j] + 1;
    }
    __syncthreads();
    for (int jj = threadIdx.y; jj < 32; jj += 8) {
        const int j = j0 + jj, q = q0 + threadIdx.x;
        if (q < nq && j < k) out[(int64_t)j * nq + q] = tile[threadIdx.x][jj];
    }
}

// host column-major [n x d] -> device row-major
double* upload_rm(bmx::DevBuf<double>& tmp, bmx::DevBuf<double>& out, const double* cm, int n, int d, hipStream_t s) {
    const double* dcm = upload(tmp, cm, (size_t)n * d, s);
    double* p = out.reserve(std::max<size_t>((size_t)n * d, 1));
    bmx::transpose_cm_to_rm(s, dcm, n, d, p);
    return p;
}

// The caller's bmx_params_t may come from an older (shorter) or a newer (longer) header: struct_size says how much of
// it is there.  Fields the caller does not have keep their defaults; bytes this library does not know must be zero.
bmx_params_t read_params(const bmx_params_t* params) {
    if (!params) throw bmx::Error(BMX_ERR_ARG, "null params");
    bmx_params_t p;
    std::memset(&p, 0, sizeof(p));
    p.sigma = 0.1;
    const int32_t sz = params->struct_size;
    const size_t need = offsetof(bmx_params_t, auto_merge) + sizeof(int32_t);
    if (sz < (int32_t)need)
        throw bmx::Error(BMX_ERR_ARG, "bmx_params_t.struct_size is not set (expected sizeof(bmx_params_t))");
    if ((size_t)sz > sizeof(p)) {
        const unsigned char* extra = reinterpret_cast<const unsigned char*>(params) + sizeof(p);
        for (size_t i = 0; i < (size_t)sz - sizeof(p); ++i)
            if (extra[i]) throw bmx::Error(BMX_ERR_ARG, "bmx_params_t carries fields this library version does not know");
    }
    std::memcpy(&p, params, std::min((size_t)sz, sizeof(p)));
    p.struct_size = (int32_t)sizeof(p);
    return p;
}

template <class T>
T* malloc_arr(size_t n) {
    T* p = (T*)std::malloc(std::max<size_t>(n, 1) * sizeof(T));
    if (!p) throw std::bad_alloc();
    return p;
}

template <class T>
T* download_malloc(const T* dev, size_t n, hipStream_t s) {
    T* p = malloc_arr<T>(n);
    if (n) {
        hipError_t e = hipMemcpyAsync(p, dev, n * sizeof(T), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            std::free(p);
            throw bmx::Error(BMX_ERR_HIP, std::string("device to host copy failed: ") + hipGetErrorString(e));
        }
    }
    return p;
}

void make_node(bmx::Engine& e, bmx::Node& node, bmx::DevBuf<double>& tmp, const double* cm, int n, int d) {
    node.index = {1};
    node.n = n;
    node.origin = {bmx::Segment{1, n}};
    upload_rm(tmp, node.data, cm, n, d, e.stream());
}

}  // namespace

extern "C" {

const char* bmx_last_error(void) { return g_last_error.c_str(); }

int32_t bmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void bmx_free(void* p) { std::free(p); }

int32_t bmx_set_device(int32_t device) {
    return guarded([&] {
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) {
            (void)hipGetLastError();  // (the runtime keeps the error for the next hipGetLastError: a later launch check would trip)
            throw bmx::Error(BMX_ERR_HIP, std::string("hipSetDevice failed: ") + hipGetErrorString(e));
        }
    });
}

void bmx_trim_caches(void) {
    // device blocks parked by engines that are gone (up to 16 GB are kept for the next engine on the device)
    bmx::DevBlockCache::release_global();
    bmx::StreamPool::release_all();  // (and the streams parked by engines that are gone)
}

int64_t bmx_last_knn_exact_fallbacks(void) { return g_last_fallbacks; }

void bmx_set_force_exact_knn(int32_t on) { g_force_exact = on; }

int32_t bmx_dev_set(const char* name, int32_t value) {
    if (!name) return BMX_ERR_ARG;
    bmx::DevKnobs& k = bmx::dev_knobs();
    const std::string n(name);
    if (n == "knn_tier") k.knn_tier = value;
    else if (n == "sample") k.sample = value;
    else if (n == "split_c") k.split_c = value;
    else if (n == "force_c") k.force_c = value;
    else if (n == "no_margin") k.no_margin = value;
    else if (n == "asv_fast") k.asv_fast = value;
    else if (n == "asv_cap") k.asv_cap = value;
    else if (n == "sample_split") k.sample_split = value;
    else if (n == "lk_seed") k.lk_seed = value;
    else if (n == "asv_modes") k.asv_modes = value;
    else if (n == "asv_sync") k.asv_sync = value;
    else if (n == "tau_replay") k.tau_replay = value;
    else if (n == "exchange_always") k.exchange_always = value;
    else if (n == "refine_wave") k.refine_wave = value;
    else if (n == "reset") k = bmx::DevKnobs();
    else {
        g_last_error = "bmx_dev_set: unknown knob '" + n + "'";
        return BMX_ERR_ARG;
    }
    return BMX_OK;
}

int32_t bmx_dev_get(const char* name, int64_t* value) {
    if (!name || !value) return BMX_ERR_ARG;
    return guarded([&] {
        const std::string n(name);
        unsigned long long t[8];
        if (n == "asv_literal_cells" || n == "asv_fallback_cells" || n == "asv_tiled_cells") {
            bmx::asv_tally_read(t, false);
            *value = (int64_t)t[n == "asv_literal_cells" ? 0 : (n == "asv_fallback_cells" ? 1 : 2)];
        } else if (n == "asv_ticks_stream" || n == "asv_ticks_wait" || n == "asv_ticks_cells") {
            bmx::asv_ticks_read(t);
            *value = (int64_t)t[n == "asv_ticks_stream" ? 0 : (n == "asv_ticks_wait" ? 1 : 2)];
        } else if (n == "asv_ticks_literal" || n == "asv_ticks_chains" || n == "asv_literal_addends" || n == "asv_chain_tiles") {
            bmx::asv_ticks_read(t);
            *value = (int64_t)t[n == "asv_ticks_literal" ? 4 : (n == "asv_ticks_chains" ? 5 : (n == "asv_literal_addends" ? 6 : 7))];
        } else if (n == "asv_tally_reset") {
            bmx::asv_tally_read(t, true);
            *value = 0;
        } else {
            throw bmx::Error(BMX_ERR_ARG, "bmx_dev_get: unknown counter '" + n + "'");
        }
    });
}

int32_t bmx_dev_get_bytes(const char* name, void* dst, int64_t n) {
    if (!name || (n > 0 && !dst) || n < 0) return BMX_ERR_ARG;
    return guarded([&] {
        if (std::string(name) != "asv_modes") throw bmx::Error(BMX_ERR_ARG, std::string("bmx_dev_get_bytes: unknown array '") + name + "'");
        bmx::asv_modes_read(static_cast<unsigned char*>(dst), (size_t)n);
    });
}

int32_t bmx_dev_host_copy(void* dst, const void* src, int64_t bytes) {
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return BMX_ERR_ARG;
    bmx::host_parallel_memcpy(dst, src, (size_t)bytes);
    return BMX_OK;
}

void bmx_shard_range(int64_t n, int32_t rank, int32_t world, int64_t* begin, int64_t* end) {
    bmx::bmx_shard_range_impl(n, rank, world, begin, end);
}

int64_t bmx_shard_gather_bytes(int64_t n, int32_t world, int64_t bytes_per_row) {
    return bmx::bmx_shard_rows_per_rank(n, world) * bytes_per_row;
}

int32_t bmx_query_knn(const double* X, int32_t nx, const double* query, int32_t nq, int32_t d, int32_t k,
                      int32_t* index, double* distance) {
    return guarded([&] {
        if (nx < 0 || nq < 0 || d <= 0 || k < 0) throw bmx::Error(BMX_ERR_ARG, "queryKNN: negative dimension");
        if (k > nx) throw bmx::Error(BMX_ERR_ARG, "queryKNN: 'k' exceeds the number of points in 'X'");
        if (nq == 0 || k == 0) return;
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> t1, t2, dX, dQ, dD;
        bmx::DevBuf<int32_t> dI, dIc;
        const double* px = upload_rm(t1, dX, X, nx, d, s);
        const double* pq = upload_rm(t2, dQ, query, nq, d, s);
        int32_t* pi = dI.reserve((size_t)nq * k);
        double* pd = dD.reserve((size_t)nq * k);
        e.knn(px, nullptr, nx, pq, nullptr, nq, k, pi, pd);
        // (k = 5 000 on 100 000 queries is 6 GB of results: no value-initialised vectors, the copies through the pinned staging
        // ring, the transposition into R's column-major layout by the pool of host threads, 512 queries a piece)
        const size_t nres = (size_t)nq * k;
        if (nres >= ((size_t)1 << 22)) {
            // large results: transposed into R's layout ON THE DEVICE and sent straight into the caller's arrays through the
            // staging ring (k = 5 000 on 100 000 queries: 6 GB that the host threads -- 16 CPUs of quota on this pool's boxes --
            // would otherwise read and write once more)
            g_last_fallbacks = e.knn_ws_.last_exact;
            if (index) {
                int32_t* pc = dIc.reserve(nres);
                hipLaunchKernelGGL(idx_to_r_layout, dim3((unsigned)((nq + 31) / 32), (unsigned)((k + 31) / 32)), dim3(32, 8), 0, s,
                                   (const int32_t*)pi, nq, k, pc);
                BMX_LAUNCH_CHECK();
                bmx::download_pageable(index, pc, nres * sizeof(int32_t), s);
            }
            if (distance) {
                bmx::DevBuf<double> dDc;
                double* pc = dDc.reserve(nres);
                bmx::transpose_rm_to_cm(s, pd, nq, k, pc, nq, 0);
                bmx::download_pageable(distance, pc, nres * sizeof(double), s);
                BMX_HIP(hipStreamSynchronize(s));
            }
            BMX_HIP(hipStreamSynchronize(s));
            return;
        }
        std::unique_ptr<int32_t[]> hi(index ? new int32_t[nres] : nullptr);
        std::unique_ptr<double[]> hd(distance ? new double[nres] : nullptr);
        if (index) bmx::download_pageable(hi.get(), pi, nres * sizeof(int32_t), s);
        if (distance) bmx::download_pageable(hd.get(), pd, nres * sizeof(double), s);
        BMX_HIP(hipStreamSynchronize(s));
        g_last_fallbacks = e.knn_ws_.last_exact;
        const int64_t piece = 512, npieces = (nq + piece - 1) / piece;
        bmx::HostPool::get().parallel_for((size_t)npieces, [&](size_t pc) {
            const int64_t q0 = (int64_t)pc * piece, q1 = std::min<int64_t>(nq, q0 + piece);
            for (int64_t j = 0; j < k; ++j)
                for (int64_t q = q0; q < q1; ++q) {
                    if (index) index[j * nq + q] = hi[(size_t)(q * k + j)] + 1;
                    if (distance) distance[j * nq + q] = hd[(size_t)(q * k + j)];
                }
        });
    });
}

int32_t bmx_find_mutual_nns(const int32_t* left, int32_t nL, int32_t k2, const int32_t* right, int32_t nR, int32_t k1,
                            int32_t** out_left, int32_t** out_right, int64_t* npairs) {
    return guarded([&] {
        if (nL < 0 || nR < 0 || k1 < 0 || k2 < 0) throw bmx::Error(BMX_ERR_ARG, "find_mutual_nns: negative dimension");
        // R layout (column-major, 1-based) -> row-major, 0-based
        std::vector<int32_t> l((size_t)nL * k2), r((size_t)nR * k1);
        for (int64_t i = 0; i < nL; ++i)
            for (int64_t j = 0; j < k2; ++j) {
                const int32_t v = left[j * nL + i];
                if (v < 1 || v > nR) throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
                l[(size_t)(i * k2 + j)] = v - 1;
            }
        for (int64_t i = 0; i < nR; ++i)
            for (int64_t j = 0; j < k1; ++j) {
                const int32_t v = right[j * nR + i];
                if (v < 1 || v > nL) throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
                r[(size_t)(i * k1 + j)] = v - 1;
            }
        bmx::Engine& e = prim(1);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<int32_t> dl, dr, cntL, offL, partR, cntR, f, sc;
        const int32_t* pl = upload(dl, l.data(), l.size(), s);
        const int32_t* pr = upload(dr, r.data(), r.size(), s);
        cntL.reserve(std::max(1, nL));
        offL.reserve((size_t)nL + 1);
        partR.reserve(std::max<size_t>(1, (size_t)nR * k1));
        cntR.reserve(std::max(1, nR));
        bmx::SortedRows sorted;
        bmx::mutual_counts(s, pl, nL, k2, pr, nR, k1, cntL.p, partR.p, cntR.p, nullptr, nullptr, nullptr, false, nullptr,
                           nullptr, &sorted);
        bmx::exclusive_scan_i32(s, e.scan_ws_, cntL.p, offL.p, nL);
        int32_t P = 0;
        BMX_HIP(hipMemcpyAsync(&P, offL.p + nL, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
        f.reserve(std::max(1, P));
        sc.reserve(std::max(1, P));
        bmx::emit_pairs(s, pl, nL, k2, pr, k1, offL.p, nullptr, nullptr, f.p, sc.p, nullptr, nullptr, &sorted);
        *out_left = download_malloc(f.p, (size_t)P, s);
        *out_right = download_malloc(sc.p, (size_t)P, s);
        *npairs = P;
    });
}

int32_t bmx_find_mutual_nn(const double* data1, int32_t n1, const double* data2, int32_t n2, int32_t d, int32_t k1,
                           int32_t k2, int32_t** first, int32_t** second, int64_t* npairs) {
    return guarded([&] {
        if (n1 < 1 || n2 < 1 || d < 1) throw bmx::Error(BMX_ERR_ARG, "findMutualNN: empty input");
        if (k1 < 1 || k2 < 1 || k1 > n1 || k2 > n2) throw bmx::Error(BMX_ERR_ARG, "findMutualNN: 'k1'/'k2' out of range");
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::Node L, R;
        bmx::DevBuf<double> t1, t2;
        make_node(e, L, t1, data1, n1, d);
        make_node(e, R, t2, data2, n2, d);
        bmx::DevBuf<int32_t> idxLR, idxRL, cntL, offL, partR, cntR, f, sc;
        idxLR.reserve((size_t)n1 * k2);
        idxRL.reserve((size_t)n2 * k1);
        e.knn(R.data.p, nullptr, n2, L.data.p, nullptr, n1, k2, idxLR.p, nullptr);
        e.knn(L.data.p, nullptr, n1, R.data.p, nullptr, n2, k1, idxRL.p, nullptr);
        cntL.reserve(n1);
        offL.reserve((size_t)n1 + 1);
        partR.reserve((size_t)n2 * k1);
        cntR.reserve(n2);
        bmx::SortedRows sorted;
        bmx::mutual_counts(s, idxLR.p, n1, k2, idxRL.p, n2, k1, cntL.p, partR.p, cntR.p, nullptr, nullptr, nullptr, false,
                           nullptr, nullptr, &sorted);
        bmx::exclusive_scan_i32(s, e.scan_ws_, cntL.p, offL.p, n1);
        int32_t P = 0;
        BMX_HIP(hipMemcpyAsync(&P, offL.p + n1, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
        f.reserve(std::max(1, P));
        sc.reserve(std::max(1, P));
        bmx::emit_pairs(s, idxLR.p, n1, k2, idxRL.p, k1, offL.p, nullptr, nullptr, f.p, sc.p, nullptr, nullptr, &sorted);
        *first = download_malloc(f.p, (size_t)P, s);
        *second = download_malloc(sc.p, (size_t)P, s);
        *npairs = P;
    });
}

int32_t bmx_mnn_average_correction(const double* refdata, int32_t n1, const double* curdata, int32_t n2, int32_t d,
                                   int32_t k1, int32_t k2, int32_t** first, int32_t** second, int64_t* npairs,
                                   double** averaged, int32_t** second_u, int32_t* U) {
    return guarded([&] {
        if (n1 < 1 || n2 < 1 || d < 1) throw bmx::Error(BMX_ERR_ARG, "empty input");
        if (k1 < 1 || k2 < 1 || k1 > n1 || k2 > n2) throw bmx::Error(BMX_ERR_ARG, "'k1'/'k2' out of range");
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::Node L, R;
        bmx::DevBuf<double> t1, t2;
        make_node(e, L, t1, refdata, n1, d);
        make_node(e, R, t2, curdata, n2, d);
        // choose_k(k, NULL, N) = k: drive find_mnn through explicit k1 / k2 by running it twice would be wasteful;
        // it takes one k, so require k1 == k2 here (the reference's own tests do, too)
        if (k1 != k2) throw bmx::Error(BMX_ERR_ARG, "this entry point needs k1 == k2");
        const bmx::Engine::MnnOut mo = e.find_mnn(L, R, k1, std::nan(""));
        bmx::DevBuf<int32_t> f, sc, srows;
        bmx::DevBuf<double> avg_cm;
        f.reserve(std::max<int64_t>(1, mo.P));
        sc.reserve(std::max<int64_t>(1, mo.P));
        bmx::emit_pairs(s, e.idxLR_.p, mo.nsel, mo.k2, e.idxRL_.p, mo.k1, e.offL_.p, nullptr, nullptr, f.p, sc.p,
                        e.lsel_.p, e.maskL_.p, &e.sorted_);
        double* avg = e.averaged_.reserve(std::max<size_t>(1, (size_t)mo.U * d));
        bmx::average_correction(s, e.red_ws_, L.data.p, nullptr, R.data.p, nullptr, d, e.second_u_.p, mo.U, e.partR_.p,
                                e.cntR_.p, mo.k1, avg);
        double* acm = avg_cm.reserve(std::max<size_t>(1, (size_t)mo.U * d));
        bmx::transpose_rm_to_cm(s, avg, mo.U, d, acm, mo.U, 0);
        *first = download_malloc(f.p, (size_t)mo.P, s);
        *second = download_malloc(sc.p, (size_t)mo.P, s);
        *npairs = mo.P;
        *averaged = download_malloc(acm, (size_t)mo.U * d, s);
        int32_t* su = download_malloc(e.second_u_.p, (size_t)mo.U, s);
        for (int i = 0; i < mo.U; ++i) su[i] += 1;
        *second_u = su;
        *U = mo.U;
    });
}

int32_t bmx_center_along_batch_vector(double* mat, int32_t n, int32_t d, const double* batch_vec,
                                      const int32_t* restrict_idx, int32_t n_restrict) {
    return guarded([&] {
        if (n < 1 || d < 1) throw bmx::Error(BMX_ERR_ARG, "empty input");
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> t, X, v, loc, cm;
        bmx::DevBuf<int32_t> rr;
        double* px = upload_rm(t, X, mat, n, d, s);
        const double* pv = upload(v, batch_vec, (size_t)d, s);
        const int32_t* pr = nullptr;
        std::vector<int32_t> z;
        if (restrict_idx && n_restrict >= 0) {
            if (n_restrict == 0) throw bmx::Error(BMX_ERR_ARG, "no cells remaining in a batch after restriction");
            z.assign(restrict_idx, restrict_idx + n_restrict);
            for (auto& x : z) {
                if (x < 1 || x > n) throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
                x -= 1;
            }
            pr = upload(rr, z.data(), z.size(), s);
        }
        // the engine's own pass: column mean over the restrict rows, then x <- x - ((x - mu) . v^) v^
        double* mu = loc.reserve((size_t)d);
        const int m = pr ? n_restrict : n;
        bmx::col_reduce(s, e.red_ws_, px, pr, 0, m, d, 0, nullptr, 1.0 / (double)m, mu);
        const int start = 0, vid = 0;
        bmx::rows_apply_stats(s, e.red_ws_, px, d, &start, &n, 1, mu, pv, &vid, 1, nullptr, nullptr, nullptr);
        double* pc = cm.reserve((size_t)n * d);
        bmx::transpose_rm_to_cm(s, px, n, d, pc, n, 0);
        BMX_HIP(hipMemcpyAsync(mat, pc, (size_t)n * d * sizeof(double), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
    });
}

int32_t bmx_tricube_weighted_correction(double* curdata, int32_t n, int32_t d, const double* correction,
                                        const int32_t* in_mnn, int32_t U, int32_t k, double ndist) {
    return guarded([&] {
        if (n < 1 || d < 1) throw bmx::Error(BMX_ERR_ARG, "empty input");
        if (U < 0 || k < 0) throw bmx::Error(BMX_ERR_ARG, "negative size");
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> t1, t2, X, C, dist, cm;
        bmx::DevBuf<int32_t> rows, idx;
        double* px = upload_rm(t1, X, curdata, n, d, s);
        const double* pc = upload_rm(t2, C, correction, U, d, s);
        std::vector<int32_t> z(in_mnn, in_mnn + U);
        for (auto& x : z) {
            if (x < 1 || x > n) throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
            x -= 1;
        }
        const int32_t* pr = upload(rows, z.data(), z.size(), s);
        const int safe_k = std::min(k, U);  // R/fastMNN.R:604
        if (safe_k > 0) {
            int32_t* pi = idx.reserve((size_t)n * safe_k);
            double* pd = dist.reserve((size_t)n * safe_k);
            e.knn(px, pr, U, px, nullptr, n, safe_k, pi, pd);
            bmx::tricube_apply(s, px, n, d, pc, pi, pd, safe_k, ndist);
        }
        double* out = cm.reserve((size_t)n * d);
        bmx::transpose_rm_to_cm(s, px, n, d, out, n, 0);
        BMX_HIP(hipMemcpyAsync(curdata, out, (size_t)n * d * sizeof(double), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
    });
}

int32_t bmx_total_variance(const double* data, int32_t n, int32_t d, double* out) {
    return guarded([&] {
        if (n < 1 || d < 1) throw bmx::Error(BMX_ERR_ARG, "empty input");
        bmx::Engine& e = prim(d);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> t, X, v;
        const double* px = upload_rm(t, X, data, n, d, s);
        // the engine's own statistics pass (one segment, slot 0): column means + total sample variance
        double* pv = v.reserve((size_t)d + 1);
        const int start = 0, slot = 0;
        bmx::rows_apply_stats(s, e.red_ws_, const_cast<double*>(px), d, &start, &n, 1, nullptr, nullptr, nullptr, 0, &slot,
                              pv + 1, pv);
        BMX_HIP(hipMemcpyAsync(out, pv, sizeof(double), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
    });
}


// HIP-event time of the kernels of the last native (.Call-level) call on this thread: bench.py's roofline of
// smooth_gaussian_kernel / adjust_shift_variance is taken over this, not over the call with its host transfers
static thread_local double g_last_native_ms = 0.0;
// (one pair of events per host thread, made at the thread's first native call and kept: no create / destroy per call)
struct NativeTimer {
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    explicit NativeTimer(hipStream_t stream) : s(stream) {
        // a call that throws, or whose events cannot be made, must not leave the PREVIOUS call's time behind
        g_last_native_ms = std::numeric_limits<double>::quiet_NaN();
        static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
        if (!ev[0] || !ev[1]) {
            for (hipEvent_t& e : ev) {
                if (e) (void)hipEventDestroy(e);
                e = nullptr;
            }
            if (hipEventCreate(&ev[0]) != hipSuccess) ev[0] = nullptr;
            if (ev[0] && hipEventCreate(&ev[1]) != hipSuccess) {
                (void)hipEventDestroy(ev[0]);
                ev[0] = ev[1] = nullptr;
            }
        }
        if (ev[0] && ev[1]) {
            a = ev[0];
            b = ev[1];
            (void)hipEventRecord(a, s);
        }
    }
    void stop() {
        if (b) (void)hipEventRecord(b, s);
    }
    void read() {  // (after the stream has been waited for)
        float ms = 0.f;
        if (a && b && hipEventElapsedTime(&ms, a, b) == hipSuccess) g_last_native_ms = ms;
    }
};
double bmx_last_native_kernel_ms(void) { return g_last_native_ms; }

int32_t bmx_smooth_gaussian_kernel(const double* averaged, int32_t g, int32_t U, const int32_t* index,
                                   int32_t index_len, const double* mat, int32_t gd, int32_t n, double sigma2,
                                   double* out) {
    return guarded([&] {
        if (U != index_len)  // src/smooth_gaussian_kernel.cpp:18-20
            throw bmx::Error(BMX_ERR_INDEX_LEN, "'index' must have length equal to number of rows in 'averaged'");
        if (g < 0 || U < 0 || gd < 0 || n < 0) throw bmx::Error(BMX_ERR_ARG, "negative dimension");
        if (n == 0 || g == 0) return;
        for (int i = 0; i < U; ++i)  // upstream reads out of bounds here; refuse instead
            if (index[i] < 0 || index[i] >= n) throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
        bmx::Engine& e = prim(1);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> dA, dM, dO, dD;
        bmx::DevBuf<int32_t> dI;
        const double* pa = upload(dA, averaged, (size_t)g * U, s);
        const double* pm = upload(dM, mat, (size_t)gd * n, s);
        const int32_t* pi = upload(dI, index, (size_t)U, s);
        double* po = dO.reserve((size_t)g * n);
        double* pd = dD.reserve((size_t)n + (size_t)std::max(1, U));  // squared norms + densities
        NativeTimer timer(s);
        bmx::smooth_gaussian_kernel_device(s, pa, g, U, pi, pm, gd, n, sigma2, po, pd);
        timer.stop();
        bmx::download_pageable(out, po, (size_t)g * n * sizeof(double), s);
        BMX_HIP(hipStreamSynchronize(s));
        timer.read();
    });
}

int32_t bmx_adjust_shift_variance(const double* data1, int32_t g1, int32_t n1, const double* data2, int32_t g2,
                                  int32_t n2, const double* vect, int32_t vrow, int32_t vcol, double sigma2,
                                  const int32_t* restrict1, int32_t nr1, const int32_t* restrict2, int32_t nr2,
                                  double* out) {
    return guarded([&] {
        if (g1 != g2 || g1 != vcol)  // src/adjust_shift_variance.cpp:33-36
            throw bmx::Error(BMX_ERR_DIM_GENES, "number of genes do not match up between matrices");
        if (n2 != vrow)  // :38-41
            throw bmx::Error(BMX_ERR_DIM_CELLS, "number of cells do not match up between matrices");
        for (int i = 0; i < nr1; ++i)  // src/utils.cpp:6-13
            if (restrict1[i] == INT32_MIN || restrict1[i] < 0 || restrict1[i] >= n1)
                throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
        for (int i = 0; i < nr2; ++i)
            if (restrict2[i] == INT32_MIN || restrict2[i] < 0 || restrict2[i] >= n2)
                throw bmx::Error(BMX_ERR_SUBSET, "subset indices out of range");
        if (n2 == 0) return;
        const int g = g1;
        bmx::Engine& e = prim(1);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> d1, d2, dv, dO, dW;
        bmx::DevBuf<int32_t> dr1, dr2;
        const double* p1 = upload(d1, data1, (size_t)g * n1, s);
        const double* p2 = upload(d2, data2, (size_t)g * n2, s);
        const double* pv = upload(dv, vect, (size_t)g * n2, s);
        const int32_t* q1 = upload(dr1, restrict1, (size_t)nr1, s);
        const int32_t* q2 = upload(dr2, restrict2, (size_t)nr2, s);
        double* po = dO.reserve(n2);
        const bmx::AsvPlan plan = bmx::adjust_shift_variance_plan(g, n2, nr1, nr2, 0);
        double* pw = dW.reserve(plan.main_doubles + plan.extra_doubles);
        NativeTimer timer(s);
        bmx::adjust_shift_variance_device(s, p1, g, n1, p2, n2, pv, sigma2, q1, nr1, q2, nr2, po, pw, plan);
        timer.stop();
        BMX_HIP(hipMemcpyAsync(out, po, (size_t)n2 * sizeof(double), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
        timer.read();
    });
}

int32_t bmx_adjust_shift_variance_form(int32_t n2, int32_t nr1, int32_t nr2) {
    return bmx::adjust_shift_variance_plan(1, n2, nr1, nr2, 1).exact ? 1 : 2;
}

int32_t bmx_cosine_norm(const double* x, int32_t G, int32_t n, double* l2, double* normalized) {
    return guarded([&] {
        if (G < 0 || n < 0) throw bmx::Error(BMX_ERR_ARG, "negative dimension");
        if (n == 0) return;
        bmx::Engine& e = prim(1);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> dx, dl, dn;
        const double* px = upload(dx, x, (size_t)G * n, s);
        double* pl = dl.reserve(n);
        bmx::cosine_l2_device(s, px, G, n, pl);
        if (l2) BMX_HIP(hipMemcpyAsync(l2, pl, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
        if (normalized && G > 0) {
            double* pn = dn.reserve((size_t)G * n);
            bmx::apply_cosine_norm_device(s, px, G, n, pl, pn);
            BMX_HIP(hipMemcpyAsync(normalized, pn, (size_t)G * n * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        BMX_HIP(hipStreamSynchronize(s));
    });
}

int32_t bmx_cosnorm_project(const double* x, int32_t G, int32_t n, const double* rotation, int32_t d,
                            const double* centers, int32_t cos_norm, double* out) {
    return guarded([&] {
        if (G < 1 || n < 0 || d < 1) throw bmx::Error(BMX_ERR_ARG, "invalid dimension");
        if (n == 0) return;
        bmx::Engine& e = prim(1);
        bmx::CacheScope cache_scope(e.cache());
        hipStream_t s = e.stream();
        bmx::DevBuf<double> dx, du, dc, dout, dcu;
        const double* px = upload(dx, x, (size_t)G * n, s);
        const double* pu = upload(du, rotation, (size_t)G * d, s);
        const double* pc = upload(dc, centers, (size_t)G, s);
        double* po = dout.reserve((size_t)n * d);
        bmx::cosnorm_project_device(s, px, G, n, pu, d, pc, cos_norm, po, nullptr, dcu.reserve(d));
        BMX_HIP(hipMemcpyAsync(out, po, (size_t)n * d * sizeof(double), hipMemcpyDeviceToHost, s));
        BMX_HIP(hipStreamSynchronize(s));
    });
}

/* ---------------------------------------------------------------- device PCA ------------------------------------ */
struct bmx_pca {
    bmx::Pca* impl = nullptr;
    ~bmx_pca() { bmx::pca_destroy(impl); }
};

int32_t bmx_pca_create(int32_t device, int32_t G, bmx_pca_t** out) {
    return guarded([&] {
        if (!out) throw bmx::Error(BMX_ERR_ARG, "null output pointer");
        if (G < 1) throw bmx::Error(BMX_ERR_ARG, "the PCA needs at least one gene");
        auto h = std::make_unique<bmx_pca>();
        h->impl = bmx::pca_create(device, G);
        *out = h.release();
    });
}

void bmx_pca_destroy(bmx_pca_t* p) { delete p; }

int32_t bmx_pca_add_batch(bmx_pca_t* p, const double* x, int64_t n, double weight, int32_t cos_norm) {
    return guarded([&] { bmx::pca_add_batch(p->impl, x, n, weight, cos_norm); });
}

int32_t bmx_pca_begin_batch(bmx_pca_t* p, int64_t n, double weight, int32_t cos_norm) {
    return guarded([&] { bmx::pca_begin_batch(p->impl, n, weight, cos_norm); });
}

int32_t bmx_pca_add_block(bmx_pca_t* p, const double* x_block, int64_t n_block) {
    return guarded([&] { bmx::pca_add_block(p->impl, x_block, n_block); });
}

int32_t bmx_pca_fit(bmx_pca_t* p, int32_t d, int32_t iters, double* centers, double* rotation, double* sdev) {
    return guarded([&] { bmx::pca_fit(p->impl, d, 0.0, iters, centers, rotation, sdev, nullptr, nullptr); });
}

int32_t bmx_pca_fit_tol(bmx_pca_t* p, int32_t d, double tol, int32_t max_iters, double* centers, double* rotation,
                        double* sdev, int32_t* iters_used, double* residual) {
    return guarded([&] {
        if (!(tol > 0.0)) throw bmx::Error(BMX_ERR_ARG, "the PCA tolerance must be positive");
        int used = 0;
        double res = 0.0;
        try {
            bmx::pca_fit(p->impl, d, tol, max_iters, centers, rotation, sdev, &used, &res);
        } catch (...) {
            if (iters_used) *iters_used = used;
            if (residual) *residual = res;
            throw;
        }
        if (iters_used) *iters_used = used;
        if (residual) *residual = res;
    });
}

int32_t bmx_pca_project(bmx_pca_t* p, int32_t batch, double* out) {
    return guarded([&] { bmx::pca_project(p->impl, batch, out); });
}

/* ---------------------------------------------------------------- engine ---------------------------------------- */

int32_t bmx_engine_create(int32_t device, bmx_engine_t** out) {
    return guarded([&] {
        if (!out) throw bmx::Error(BMX_ERR_ARG, "null output pointer");
        auto h = std::make_unique<bmx_engine>();
        h->impl = std::make_unique<bmx::Engine>(device);
        *out = h.release();
    });
}

void bmx_engine_destroy(bmx_engine_t* e) { delete e; }

int32_t bmx_engine_set_shard(bmx_engine_t* e, int32_t rank, int32_t world, bmx_allgather_fn fn, void* ctx) {
    return guarded([&] { e->impl->set_shard(rank, world, fn, ctx); });
}

int32_t bmx_rccl_load(const char* librccl_path) {
    return guarded([&] { bmx::rccl::load(librccl_path); });
}

int32_t bmx_rccl_unique_id(void* id_out, int32_t bytes) {
    return guarded([&] {
        if (!id_out || bytes < 128) throw bmx::Error(BMX_ERR_ARG, "the RCCL unique id needs 128 bytes");
        if (!bmx::rccl::api().ready()) throw bmx::Error(BMX_ERR_EXCHANGE, "RCCL is not loaded (bmx_rccl_load)");
        bmx::rccl::UniqueId id;
        const int rc = bmx::rccl::api().GetUniqueId(&id);
        if (rc != 0) throw bmx::Error(BMX_ERR_EXCHANGE, "ncclGetUniqueId failed");
        std::memcpy(id_out, id.internal, 128);
    });
}

int32_t bmx_engine_init_rccl(bmx_engine_t* e, int32_t rank, int32_t world, const void* unique_id, int32_t bytes) {
    return guarded([&] {
        if (bytes < 128) throw bmx::Error(BMX_ERR_ARG, "the RCCL unique id needs 128 bytes");
        e->impl->init_rccl(rank, world, unique_id);
    });
}

int32_t bmx_engine_emulate(bmx_engine_t* e, int32_t mode, int32_t rank, int32_t world) {
    return guarded([&] { e->impl->emulate(mode, rank, world); });
}

int32_t bmx_engine_exchange_stats(bmx_engine_t* e, int64_t* calls, int64_t* bytes) {
    return guarded([&] {
        if (calls) *calls = e->impl->exchange_calls();
        if (bytes) *bytes = e->impl->exchange_bytes();
    });
}

int32_t bmx_engine_upload(bmx_engine_t* e, int32_t nbatches, int32_t d, const double* const* data,
                          const int32_t* nrows, const int32_t* const* restrict_idx, const int32_t* n_restrict) {
    return guarded([&] { e->impl->upload(nbatches, d, data, nrows, restrict_idx, n_restrict); });
}

int32_t bmx_engine_run(bmx_engine_t* e, const bmx_params_t* params, const int32_t* tree, int32_t tree_len) {
    return guarded([&] {
        const bmx_params_t p = read_params(params);
        e->impl->knn_ws_.force_exact = g_force_exact;
        e->impl->run(p, tree, tree_len);
    });
}

int32_t bmx_engine_download(bmx_engine_t* e, double* corrected, int32_t* batch, int32_t* merge_left,
                            int32_t* merge_right, double* batch_size, int32_t* skipped, double* lost_var) {
    return guarded([&] { e->impl->download(corrected, batch, merge_left, merge_right, batch_size, skipped, lost_var); });
}

int32_t bmx_engine_pairs(bmx_engine_t* e, int32_t merge, int32_t** left, int32_t** right, int64_t* npairs) {
    return guarded([&] { e->impl->pairs(merge, left, right, npairs); });
}

int32_t bmx_engine_pairs_into(bmx_engine_t* e, int32_t merge, int32_t* left, int32_t* right, int64_t capacity,
                              int64_t* npairs) {
    return guarded([&] {
        const int64_t P = e->impl->pairs_count(merge);
        if (npairs) *npairs = P;
        if (!left || !right) return;  // size query
        if (capacity < P) throw bmx::Error(BMX_ERR_ARG, "bmx_engine_pairs_into: the arrays are too short");
        e->impl->pairs_into(merge, left, right);
    });
}

int32_t bmx_engine_pairs_all_into(bmx_engine_t* e, int32_t nmerges, int32_t* const* left, int32_t* const* right,
                                  const int64_t* capacity) {
    return guarded([&] {
        if (!left || !right || !capacity) throw bmx::Error(BMX_ERR_ARG, "bmx_engine_pairs_all_into: null argument");
        e->impl->pairs_all_into(nmerges, left, right, capacity);
    });
}

int32_t bmx_engine_merge_stats(bmx_engine_t* e, int32_t merge, int64_t* out6) {
    return guarded([&] { e->impl->merge_stats(merge, out6); });
}

int32_t bmx_engine_set_watchdog(bmx_engine_t* e, double base_ms) {
    return guarded([&] { e->impl->set_watchdog(1e-3 * base_ms); });
}

int32_t bmx_engine_debug_stall(bmx_engine_t* e, int32_t ms) {
    return guarded([&] { e->impl->debug_stall(ms); });
}

int32_t bmx_engine_set_profiling(bmx_engine_t* e, int32_t on) {
    return guarded([&] { e->impl->set_profiling(on != 0); });
}

int32_t bmx_engine_profile(bmx_engine_t* e, double* topk_ms, int64_t* topk_launches, int64_t* exact_fallbacks) {
    return guarded([&] { e->impl->profile(topk_ms, topk_launches, exact_fallbacks); });
}

int32_t bmx_engine_set_snapshot(bmx_engine_t* e, int32_t merge) {
    return guarded([&] { e->impl->set_snapshot(merge); });
}

int32_t bmx_engine_snapshot(bmx_engine_t* e, double* left_rm, double* right_rm, int64_t* n_left, int64_t* n_right) {
    return guarded([&] { e->impl->snapshot(left_rm, right_rm, n_left, n_right); });
}

int32_t bmx_engine_snapshot_var_adj(bmx_engine_t* e, double* left_rm, double* right_rm, double* corr_rm, double* scaling,
                                    int32_t* restrict1, int32_t* restrict2, int64_t* sizes4) {
    return guarded([&] { e->impl->snapshot_var_adj(left_rm, right_rm, corr_rm, scaling, restrict1, restrict2, sizes4); });
}

int32_t bmx_engine_var_adj_tally(bmx_engine_t* e, int32_t merge, int64_t* out3) {
    return guarded([&] {
        if (!out3) throw bmx::Error(BMX_ERR_ARG, "null output");
        e->impl->var_adj_tally(merge, out3);
    });
}

int32_t bmx_engine_snapshot_var_adj_modes(bmx_engine_t* e, uint8_t* dst, int64_t n) {
    return guarded([&] {
        if (n < 0 || (n > 0 && !dst)) throw bmx::Error(BMX_ERR_ARG, "null output");
        e->impl->snapshot_var_adj_modes(dst, n);
    });
}

int32_t bmx_engine_profile_var_adj(bmx_engine_t* e, double* out3) {
    return guarded([&] { e->impl->profile_var_adj(out3); });
}

int32_t bmx_engine_profile_detail(bmx_engine_t* e, double* out10) {
    return guarded([&] { e->impl->profile_detail(out10); });
}

int32_t bmx_engine_knn_kernel(bmx_engine_t* e, char* buf, int32_t n) {
    return guarded([&] {
        if (!buf || n < 1) throw bmx::Error(BMX_ERR_ARG, "no buffer");
        const std::string& k = e->impl->knn_ws_.last_kernel;
        std::strncpy(buf, k.c_str(), (size_t)n - 1);
        buf[n - 1] = 0;
    });
}

int32_t bmx_engine_knn_variant(bmx_engine_t* e) { return e && e->impl ? e->impl->knn_ws_.last_variant : -1; }

int32_t bmx_fast_mnn(int32_t nbatches, int32_t d, const double* const* data, const int32_t* nrows,
                     const int32_t* const* restrict_idx, const int32_t* n_restrict, const bmx_params_t* params,
                     const int32_t* tree, int32_t tree_len, double* corrected, int32_t* batch, int32_t* merge_left,
                     int32_t* merge_right, double* batch_size, int32_t* skipped, double* lost_var,
                     bmx_engine_t** out_engine) {
    return guarded([&] {
        const bmx_params_t p = read_params(params);
        int dev = 0;
        BMX_HIP(hipGetDevice(&dev));
        auto h = std::make_unique<bmx_engine>();
        h->impl = std::make_unique<bmx::Engine>(dev);
        h->impl->knn_ws_.force_exact = g_force_exact;
        // the caller's matrices stay valid for the whole call: their upload is pulled by the run itself, batch by batch
        // (the first two ahead of merge 1, the others while the GPU is busy searching)
        const auto t0 = std::chrono::steady_clock::now();
        // The result matrix is the caller's fresh allocation (R's allocMatrix, numpy's empty): 320 MB at config 3 = 78 000
        // pages that fault on first touch, and the download's host copy was bound by exactly that (10 ms at 32 GB/s).  The
        // host has nothing to do while the GPU runs: a few short-lived threads touch every page now (one byte each; the
        // library overwrites every element before it returns), the download then copies into resident pages.
        // (The size of that matrix comes from the caller's nbatches / nrows / d: nothing is written before they have been
        // checked the way upload() checks them -- a wrong count must be BMX_ERR_ARG, not a dozen threads writing through wild
        // memory -- and the joiner exists before the first thread does, so a failed thread creation unwinds cleanly and the
        // call goes on without the touching.)
        std::vector<std::thread> toucher;
        struct Joiner {
            std::vector<std::thread>& v;
            ~Joiner() {
                for (auto& t : v)
                    if (t.joinable()) t.join();
            }
        } joiner{toucher};
        bool sizes_ok = corrected && nbatches >= 2 && nbatches <= 65536 && d >= 1 && nrows && data;
        int64_t N = 0;
        for (int b = 0; sizes_ok && b < nbatches; ++b) {
            sizes_ok = nrows[b] >= 0 && data[b] != nullptr;
            N += nrows[b];
        }
        if (sizes_ok && N <= (int64_t)2147483647) {
            const size_t bytes = (size_t)N * (size_t)d * sizeof(double);
            if (bytes >= ((size_t)8 << 20)) {
                constexpr int nt = 12;
                volatile char* base = reinterpret_cast<volatile char*>(corrected);
                try {
                    toucher.reserve(nt);
                    for (int t = 0; t < nt; ++t)
                        toucher.emplace_back([base, bytes, t] {
                            const size_t lo = bytes / nt * t, hi = t + 1 == nt ? bytes : bytes / nt * (t + 1);
                            for (size_t o = lo; o < hi; o += 4096) base[o] = 0;
                        });
                } catch (const std::system_error&) {  // (no more threads to be had: the download faults the pages itself)
                }
            }
        }
        h->impl->upload(nbatches, d, data, nrows, restrict_idx, n_restrict, /* lazy */ true);
        const auto t1 = std::chrono::steady_clock::now();
        h->impl->run(p, tree, tree_len);
        const auto t2 = std::chrono::steady_clock::now();
        for (auto& t : toucher)
            if (t.joinable()) t.join();
        h->impl->download(corrected, batch, merge_left, merge_right, batch_size, skipped, lost_var);
        if (bmx::debug_timings()) {
            const auto t3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[bmx] one-shot: create+upload %.2f ms, run %.2f ms, download %.2f ms; %ld hipMallocs so far\n", ms(t0, t1),
                    ms(t1, t2), ms(t2, t3), bmx::dev_malloc_calls().load());
        }
        if (out_engine) *out_engine = h.release();
    });
}

}  // extern "C"
