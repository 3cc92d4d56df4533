// Selection machinery shared by the MFMA top-k kernels (f32 and split-bf16 candidate passes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace bmx {
namespace sel {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t f32_orderable(float v) {
    uint32_t u = __float_as_uint(v);
    return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float orderable_f32(uint32_t o) {
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xFFFFFFFFu);
    return __uint_as_float(u);
}

// ---- error bound of a candidate pass (shared by the prep kernels, which hand the pass its margins and seeded thresholds,
// and by knn_refine, which certifies with the same numbers) ----------------------------------------------------------
struct PassEps {
    double eps_k, eps_qr, eps_split, eps_den;
    int scaled;  // fp16 tier: coordinates times the power of two pass_scale(max |r|^2)
};

// power-of-two scale that puts the largest reference norm into (24, 48]; 1 for degenerate input (= f16_scale, knn_f16.hip)
__host__ __device__ __forceinline__ double pass_scale(double max_n2) {
    const double rm = sqrt(max_n2);
    if (!(rm > 1e-150) || !(rm < 1e150)) return 1.0;
    int e = 0;
    (void)frexp(48.0 / rm, &e);
    return ldexp(1.0, e - 1);
}

// error bound of a candidate pass for one query (unscaled units): f32 rounding of the centred coordinates +
// accumulation over eps_k terms, the low-order products the pass drops (eps_split |q||r|), and fp16 inputs below the
// normal range taken as flushed to zero (eps_den, scaled units)
__device__ __forceinline__ double pass_eps(double qn, double rm, double s, double eps_k, double eps_qr, double eps_split,
                                           double eps_den) {
    const double u = 5.9604644775390625e-8;  // 2^-24
    return 1.5 * u * (2.0 * (qn + rm) * (qn + rm) + (eps_k + 1.0) * (rm * rm + eps_qr * qn * rm)) +
           eps_split * qn * rm + eps_den * ((2.0 * qn + rm) * s + 1.0) / (s * s);
}

// Twice the pass's error bound for a query of squared norm qn2, in the pass's own (scaled) units, rounded up: a reference
// whose approximate value exceeds the k-th best approximate value by more than this is farther, exactly, than each of
// those k -- the cut knn_refine applies to the candidates, handed to the pass itself so that it stops collecting them.
__device__ __forceinline__ float pass_margin(double qn2, double max_rn2, const PassEps& pe) {
    const double s = pe.scaled ? pass_scale(max_rn2) : 1.0;
    const double eps = pass_eps(sqrt(qn2), sqrt(max_rn2), s, pe.eps_k, pe.eps_qr, pe.eps_split, pe.eps_den);
    const double x = 2.0 * eps * (s * s) * 1.0000002 + 1e-30;
    return (float)(x * 1.000001);  // (the f32 rounding cannot land below x)
}

// Seeded search: the starting threshold of a query in the pass's own units, as its order-preserving image.  A reference
// within seed_d2 of the query has an approximate value below (seed - |q~|^2 + eps) s^2, so nothing the caller cares about is
// filtered out.
__device__ __forceinline__ uint32_t pass_seed_tau(double seed_d2, double qn2, double max_rn2, const PassEps& pe) {
    const double s = pe.scaled ? pass_scale(max_rn2) : 1.0;
    const double eps = pass_eps(sqrt(qn2), sqrt(max_rn2), s, pe.eps_k, pe.eps_qr, pe.eps_split, pe.eps_den);
    const double x = (seed_d2 - qn2 + eps) * (s * s);
    const float t = (float)(x + fabs(x) * 9.5367431640625e-7 + 1e-30);  // + 2^-20 relative: the f32 rounding cannot land below x
    return f32_orderable(t);
}

// Per query slot the LDS holds KS "kept" entries (sorted, shared by the two lanes that own the query's two K-halves)
// followed by two lane-private pending lists of PL entries each: a lane appends with a plain ds_write (no atomics, no
// returned value to wait for).  When a pending list fills, one wave merges kept + both pending lists by rank-counting
// over the unique 64-bit keys (orderable value << 32 | reference index), keeps the KS smallest and tightens tau.
constexpr int PL = 12;
// Row pitch of a query's list in 8-byte entries (register-state variant): one more than the capacity, so that the
// lists of neighbouring queries start 2 banks apart and the appends of a wave's lanes do not pile up on one bank pair.
__host__ __device__ constexpr int list_pitch(int KS, int PLN) { return KS + 2 * PLN + 1; }

template <int KS>
__device__ __forceinline__ void compact_slot(unsigned long long* buf, int* kcnt, float* tau_s, int slot, int jj,
                                             int lane, int& mycnt) {
    constexpr int CAP = KS + 2 * PL;
    const int nk = kcnt[slot];
    const int n0 = __builtin_amdgcn_readlane(mycnt, jj);
    const int n1 = __builtin_amdgcn_readlane(mycnt, jj + 32);
    const int n = nk + n0 + n1;
    unsigned long long* b = buf + slot * CAP;
    int src = lane;  // kept entries sit at [0, nk)
    if (lane >= nk) src = lane < nk + n0 ? KS + (lane - nk) : KS + PL + (lane - nk - n0);
    const unsigned long long key = lane < n ? b[src] : ~0ull;
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    int rank = 0;
    for (int f = 0; f < n; ++f) {
        const uint32_t flo = __builtin_amdgcn_readlane(klo, f);
        const uint32_t fhi = __builtin_amdgcn_readlane(khi, f);
        const unsigned long long fk = ((unsigned long long)fhi << 32) | flo;
        rank += fk < key ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its key before slots are rewritten
    if (lane < n && rank < KS) b[rank] = key;
    if (n >= KS && lane < n && rank == KS - 1) tau_s[slot] = orderable_f32(khi);
    if (lane == 0) kcnt[slot] = n < KS ? n : KS;
    if (lane == jj || lane == jj + 32) mycnt = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Register-state variant used by the split-bf16 ring kernel: the number of kept entries (nk) and the working
// threshold (tau) of a query live in the registers of the two lanes that own it, so a compaction makes ONE LDS round
// trip (the entries).  Entries sit in the LDS as (raw f32 bits << 32 | reference index).
// Ranking uses a 31-bit key per lane: the order-preserving integer image of the value, shifted right by one, with
// its low 6 bits replaced by the lane number -- unique, and a comparison is the sign of a difference; entries are
// broadcast by readlane in 8-entry chunks with constant lane numbers.
// Values closer than 2^-16 relative may therefore swap places at the cut; the new threshold is the cut key with the
// lane bits cleared, which is <= the value of everything dropped, so "rejected => value >= tau" still holds exactly.
template <int KS, int PLN>
__device__ __forceinline__ void compact_regs(unsigned long long* buf, int slot, int jj, int lane, int& mycnt, int& nk_reg,
                                             float& tau) {
    constexpr int CAP = KS + 2 * PLN;
    const int nk = __builtin_amdgcn_readlane(nk_reg, jj);
    const int n0 = __builtin_amdgcn_readlane(mycnt, jj);
    const int n1 = __builtin_amdgcn_readlane(mycnt, jj + 32);
    const int n = nk + n0 + n1;
    unsigned long long* b = buf + slot * list_pitch(KS, PLN);
    int src = lane;  // kept entries sit at [0, nk)
    if (lane >= nk) src = lane < nk + n0 ? KS + (lane - nk) : KS + PLN + (lane - nk - n0);
    const unsigned long long raw = lane < n ? b[src] : 0ull;
    const uint32_t key =
        lane < n ? (((f32_orderable(__uint_as_float((uint32_t)(raw >> 32))) >> 1) & ~63u) | (uint32_t)lane) : 0x7FFFFFFFu;
    int rank = 0;
#pragma unroll
    for (int c = 0; c < (CAP + 7) / 8; ++c) {
        if (c * 8 < n) {
            // 31-bit keys: the sign of the difference is the comparison, so no lane mask (and no scalar-register
            // round trip per entry) is involved
            uint32_t fk[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) fk[e] = c * 8 + e < CAP ? (uint32_t)__builtin_amdgcn_readlane(key, c * 8 + e) : 0x7FFFFFFFu;
#pragma unroll
            for (int e = 0; e < 8; ++e) rank += (fk[e] - key) >> 31;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // every lane holds its entry before slots are rewritten
    if (lane < n && rank < KS) b[rank] = raw;
    const bool mine = lane == jj || lane == jj + 32;
    if (n >= KS) {
        const unsigned long long at = __builtin_amdgcn_ballot_w64(lane < n && rank == KS - 1);
        const float kth = orderable_f32(((uint32_t)__builtin_amdgcn_readlane(key, __builtin_ctzll(at)) & ~63u) << 1);
        if (mine) tau = kth < tau ? kth : tau;
    }
    if (mine) {
        mycnt = 0;
        nk_reg = n < KS ? n : KS;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

}  // namespace sel
}  // namespace bmx
