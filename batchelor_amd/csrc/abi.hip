// extern "C" boundary of libbatchelor_mi355x.so (declared in include/batchelor_mi355x.h).
// Nothing throws across it: exceptions become return codes + a thread-local message.
#include <cstdlib>
#include <cstring>
#include <functional>

#include "bmx_common.hpp"
#include "bmx_ops.hpp"

namespace {

thread_local std::string g_last_error;
thread_local int64_t g_last_fallbacks = 0;
thread_local int g_force_exact = 0;

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

struct Stream {
    hipStream_t s = nullptr;
    Stream() { BMX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~Stream() {
        if (s) (void)hipStreamDestroy(s);
    }
};

// column-major [n x d] (R) <-> row-major [n x d] on the host; the engine path transposes on the device instead
std::vector<double> to_row_major(const double* cm, int64_t n, int64_t d) {
    std::vector<double> out((size_t)(n * d));
    for (int64_t c = 0; c < d; ++c)
        for (int64_t r = 0; r < n; ++r) out[(size_t)(r * d + c)] = cm[c * n + r];
    return out;
}

template <class T>
T* upload(bmx::DevBuf<T>& buf, const T* host, size_t n, hipStream_t s) {
    T* p = buf.reserve(std::max<size_t>(n, 1));
    if (n) BMX_HIP(hipMemcpyAsync(p, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    return p;
}

int32_t* malloc_i32(size_t n) {
    int32_t* p = (int32_t*)std::malloc(std::max<size_t>(n, 1) * sizeof(int32_t));
    if (!p) throw std::bad_alloc();
    return p;
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

int64_t bmx_last_knn_exact_fallbacks(void) { return g_last_fallbacks; }

void bmx_set_force_exact_knn(int32_t on) { g_force_exact = on; }

int32_t bmx_query_knn(const double* X, int32_t nx, const double* query, int32_t nq, int32_t d, int32_t k,
                      int32_t* index, double* distance) {
    return guarded([&] {
        if (nx < 0 || nq < 0 || d <= 0 || k < 0) throw bmx::Error(BMX_ERR_ARG, "queryKNN: negative dimension");
        if (k > nx) throw bmx::Error(BMX_ERR_ARG, "queryKNN: 'k' exceeds the number of points in 'X'");
        if (nq == 0 || k == 0) return;
        Stream st;
        bmx::KnnWorkspace ws;
        ws.force_exact = g_force_exact;
        bmx::DevBuf<double> dX, dQ, dD;
        bmx::DevBuf<int32_t> dI;
        auto hx = to_row_major(X, nx, d);
        auto hq = to_row_major(query, nq, d);
        const double* px = upload(dX, hx.data(), hx.size(), st.s);
        const double* pq = upload(dQ, hq.data(), hq.size(), st.s);
        int32_t* pi = dI.reserve((size_t)nq * k);
        double* pd = dD.reserve((size_t)nq * k);
        bmx::knn_device(st.s, ws, px, nullptr, nx, pq, nullptr, nq, d, k, pi, pd, 0, nq);
        std::vector<int32_t> hi((size_t)nq * k);
        std::vector<double> hd((size_t)nq * k);
        int32_t nflag = 0;
        BMX_HIP(hipMemcpyAsync(hi.data(), pi, hi.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st.s));
        BMX_HIP(hipMemcpyAsync(hd.data(), pd, hd.size() * sizeof(double), hipMemcpyDeviceToHost, st.s));
        BMX_HIP(hipMemcpyAsync(&nflag, ws.flagged.p, sizeof(int32_t), hipMemcpyDeviceToHost, st.s));
        BMX_HIP(hipStreamSynchronize(st.s));
        g_last_fallbacks = nflag;
        for (int64_t q = 0; q < nq; ++q)
            for (int64_t j = 0; j < k; ++j) {
                if (index) index[j * nq + q] = hi[(size_t)(q * k + j)] + 1;
                if (distance) distance[j * nq + q] = hd[(size_t)(q * k + j)];
            }
    });
}

}  // extern "C"
